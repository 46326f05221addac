// K1 instantiations: n-D integrator "quad-toy" -- BASELINE.json configs[2] is (nx,nu)=(16,12).
// One translation unit per group of dimensions (MPPI_INTEGRATOR_GROUP, set by _build.py): hipcc runs the units
// in parallel and the (16,12) kernels -- rollout, single-launch command, multi-rollout, KMPPI-fused -- are
// as much work for the compiler as all the small ones together.
#include "dispatch.hpp"
#include "rollout.hpp"
#ifndef MPPI_INTEGRATOR_GROUP
#define MPPI_INTEGRATOR_GROUP 0
#endif
namespace mppi {
#define MPPI_INTEGRATOR_DIMS_A(X) X(16, 12)
#define MPPI_INTEGRATOR_DIMS_B(X) X(6, 4) X(2, 2) X(4, 2) X(8, 4) X(12, 6)
#if MPPI_INTEGRATOR_GROUP == 0
#define MPPI_INTEGRATOR_DIMS(X) MPPI_INTEGRATOR_DIMS_A(X)
int rollout_integrator_small(const KArgs<float>& a, hipStream_t st);
int rollout_integrator_small(const KArgs<double>& a, hipStream_t st);
bool supported_integrator(int nx, int nu, int) {
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_INTEGRATOR_DIMS_A(X) MPPI_INTEGRATOR_DIMS_B(X)
#undef X
  return false;
}
#else
#define MPPI_INTEGRATOR_DIMS(X) MPPI_INTEGRATOR_DIMS_B(X)
#endif
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<IntegratorModel<T, NX, NU>, T>(a, st);
  MPPI_INTEGRATOR_DIMS(X)
#undef X
#if MPPI_INTEGRATOR_GROUP == 0
  return rollout_integrator_small(a, st);
#else
  return MPPI_E_UNSUPPORTED;
#endif
}
#if MPPI_INTEGRATOR_GROUP == 0
int rollout_integrator(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_integrator(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
#else
int rollout_integrator_small(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_integrator_small(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
#endif
}  // namespace mppi
