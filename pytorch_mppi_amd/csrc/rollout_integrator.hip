// K1 instantiations: n-D integrator "quad-toy" -- BASELINE.json configs[2] is (nx,nu)=(16,12).
// One translation unit per group of dimensions (MPPI_INTEGRATOR_GROUP, set by _build.py): hipcc runs the units in
// parallel, and every (nx,nu) pair carries the whole kernel family -- rollout, single-launch command, multi-rollout,
// KMPPI-fused, on-chip -- so a unit with five pairs was a four-minute compile.  Group g handles its pairs and hands
// everything else to group g + 1.
#include "dispatch.hpp"
#include "rollout.hpp"
#ifndef MPPI_INTEGRATOR_GROUP
#define MPPI_INTEGRATOR_GROUP 0
#endif
namespace mppi {
#define MPPI_INTEGRATOR_DIMS_0(X) X(16, 12)
#define MPPI_INTEGRATOR_DIMS_1(X) X(6, 4) X(2, 2)
#define MPPI_INTEGRATOR_DIMS_2(X) X(4, 2) X(8, 4)
#define MPPI_INTEGRATOR_DIMS_3(X) X(12, 6)
#define MPPI_CAT_(a, b) a##b
#define MPPI_CAT(a, b) MPPI_CAT_(a, b)
#if MPPI_INTEGRATOR_GROUP == 0
#define MPPI_THIS rollout_integrator
#define MPPI_NEXT rollout_integrator_g1
#define MPPI_DIMS MPPI_INTEGRATOR_DIMS_0
bool supported_integrator(int nx, int nu, int) {
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_INTEGRATOR_DIMS_0(X) MPPI_INTEGRATOR_DIMS_1(X) MPPI_INTEGRATOR_DIMS_2(X) MPPI_INTEGRATOR_DIMS_3(X)
#undef X
  return false;
}
#elif MPPI_INTEGRATOR_GROUP == 1
#define MPPI_THIS rollout_integrator_g1
#define MPPI_NEXT rollout_integrator_g2
#define MPPI_DIMS MPPI_INTEGRATOR_DIMS_1
#elif MPPI_INTEGRATOR_GROUP == 2
#define MPPI_THIS rollout_integrator_g2
#define MPPI_NEXT rollout_integrator_g3
#define MPPI_DIMS MPPI_INTEGRATOR_DIMS_2
#else
#define MPPI_THIS rollout_integrator_g3
#define MPPI_DIMS MPPI_INTEGRATOR_DIMS_3
#endif
#ifdef MPPI_NEXT
int MPPI_NEXT(const KArgs<float>& a, hipStream_t st);
int MPPI_NEXT(const KArgs<double>& a, hipStream_t st);
#endif
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<IntegratorModel<T, NX, NU>, T>(a, st);
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
