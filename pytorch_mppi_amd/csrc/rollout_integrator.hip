// K1 instantiations: n-D integrator "quad-toy" -- BASELINE.json configs[2] is (nx,nu)=(16,12)
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
#define MPPI_INTEGRATOR_DIMS(X) X(16, 12) X(6, 4) X(2, 2) X(4, 2) X(8, 4) X(12, 6)
bool supported_integrator(int nx, int nu, int) {
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_INTEGRATOR_DIMS(X)
#undef X
  return false;
}
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<IntegratorModel<T, NX, NU>, T>(a, st);
  MPPI_INTEGRATOR_DIMS(X)
#undef X
  return MPPI_E_UNSUPPORTED;
}
int rollout_integrator(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_integrator(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
}  // namespace mppi
