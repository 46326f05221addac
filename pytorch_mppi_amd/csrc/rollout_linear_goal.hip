// K1 instantiations: linear dynamics + quadratic goal cost (reference tests/test_mppi.py:25-51).
// Two translation units (MPPI_LINEAR_GROUP, set by _build.py) so that hipcc compiles them in parallel.
#include "dispatch.hpp"
#include "rollout.hpp"
#ifndef MPPI_LINEAR_GROUP
#define MPPI_LINEAR_GROUP 0
#endif
namespace mppi {
#define MPPI_LINEAR_DIMS_A(X) X(12, 4) X(2, 2)
#define MPPI_LINEAR_DIMS_B(X) X(4, 2) X(6, 3) X(10, 3)
#if MPPI_LINEAR_GROUP == 0
#define MPPI_LINEAR_DIMS(X) MPPI_LINEAR_DIMS_A(X)
int rollout_linear_goal_b(const KArgs<float>& a, hipStream_t st);
int rollout_linear_goal_b(const KArgs<double>& a, hipStream_t st);
bool supported_linear_goal(int nx, int nu, int) {
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_LINEAR_DIMS_A(X) MPPI_LINEAR_DIMS_B(X)
#undef X
  return false;
}
#else
#define MPPI_LINEAR_DIMS(X) MPPI_LINEAR_DIMS_B(X)
#endif
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
  if (a.mp == nullptr) return MPPI_E_BADARG;
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<LinearGoalModel<T, NX, NU>, T>(a, st);
  MPPI_LINEAR_DIMS(X)
#undef X
#if MPPI_LINEAR_GROUP == 0
  return rollout_linear_goal_b(a, st);
#else
  return MPPI_E_UNSUPPORTED;
#endif
}
#if MPPI_LINEAR_GROUP == 0
int rollout_linear_goal(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_linear_goal(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
#else
int rollout_linear_goal_b(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int rollout_linear_goal_b(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
#endif
}  // namespace mppi
