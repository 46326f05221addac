// K1 instantiations: pendulum (nx=2, nu=1) -- BASELINE.json configs[0..1]
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
bool supported_pendulum(int nx, int nu, int) { return nx == 2 && nu == 1; }
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
  if (!supported_pendulum(a.nx, a.nu, 0)) return MPPI_E_UNSUPPORTED;
  return launch_rollout<PendulumModel<T>, T>(a, st);
}
int rollout_pendulum(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_pendulum(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
}  // namespace mppi
