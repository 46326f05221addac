// K1 instantiations: linear dynamics + quadratic goal cost (reference tests/test_mppi.py:25-51)
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
#define MPPI_LINEAR_DIMS(X) X(2, 2) X(4, 2) X(6, 3) X(10, 3) X(12, 4)
bool supported_linear_goal(int nx, int nu, int) {
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_LINEAR_DIMS(X)
#undef X
  return false;
}
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
  if (a.mp == nullptr) return MPPI_E_BADARG;
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<LinearGoalModel<T, NX, NU>, T>(a, st);
  MPPI_LINEAR_DIMS(X)
#undef X
  return MPPI_E_UNSUPPORTED;
}
int rollout_linear_goal(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_linear_goal(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
}  // namespace mppi
