// K1 instantiations: linear dynamics + quadratic goal cost (reference tests/test_mppi.py:25-51).
// One translation unit per group of dimensions (MPPI_LINEAR_GROUP, set by _build.py), chained like rollout_integrator.hip.
#include "dispatch.hpp"
#include "rollout.hpp"
#ifndef MPPI_LINEAR_GROUP
#define MPPI_LINEAR_GROUP 0
#endif
namespace mppi {
#define MPPI_LINEAR_DIMS_0(X) X(12, 4)
#define MPPI_LINEAR_DIMS_1(X) X(2, 2) X(4, 2)
#define MPPI_LINEAR_DIMS_2(X) X(6, 3) X(10, 3)
#if MPPI_LINEAR_GROUP == 0
#define MPPI_THIS rollout_linear_goal
#define MPPI_NEXT rollout_linear_goal_g1
#define MPPI_DIMS MPPI_LINEAR_DIMS_0
bool supported_linear_goal(int nx, int nu, int) {
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_LINEAR_DIMS_0(X) MPPI_LINEAR_DIMS_1(X) MPPI_LINEAR_DIMS_2(X)
#undef X
  return false;
}
#elif MPPI_LINEAR_GROUP == 1
#define MPPI_THIS rollout_linear_goal_g1
#define MPPI_NEXT rollout_linear_goal_g2
#define MPPI_DIMS MPPI_LINEAR_DIMS_1
#else
#define MPPI_THIS rollout_linear_goal_g2
#define MPPI_DIMS MPPI_LINEAR_DIMS_2
#endif
#ifdef MPPI_NEXT
int MPPI_NEXT(const KArgs<float>& a, hipStream_t st);
int MPPI_NEXT(const KArgs<double>& a, hipStream_t st);
#endif
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
  if (a.mp == nullptr) return MPPI_E_BADARG;
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<LinearGoalModel<T, NX, NU>, T>(a, st);
  MPPI_DIMS(X)
#undef X
#ifdef MPPI_NEXT
  return MPPI_NEXT(a, st);
#else
  return MPPI_E_UNSUPPORTED;
#endif
}
int MPPI_THIS(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int MPPI_THIS(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
}  // namespace mppi
