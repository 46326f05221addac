// K1 instantiations: 2-layer MLP residual dynamics, per-lane VALU form (any hidden width, f32/f64).
// BASELINE.json configs[3..4] are (nx,nu,H)=(16,4,256).
#include <cstdlib>
#include "dispatch.hpp"
#include "rollout.hpp"
namespace mppi {
// One translation unit per (nx,nu) pair (MPPI_MLP_GROUP, set by _build.py; chained like rollout_integrator.hip): the per-lane
// MLP with its runtime hidden width was a 224-second compile as one unit, the longest of the build.
#ifndef MPPI_MLP_GROUP
#define MPPI_MLP_GROUP 0
#endif
#define MPPI_MLP_DIMS_0(X) X(16, 4)
#define MPPI_MLP_DIMS_1(X) X(2, 1)
#define MPPI_MLP_DIMS_2(X) X(4, 2)
// the further shapes of the split-operand matrix-core kernel (rollout_mlp_split.hip MPPI_SPLIT_DIMS_1): the per-lane form is what
// runs them in fp64, with M > 1 rollouts, with the visited states asked for, and with rows that kernel does not read
#define MPPI_MLP_DIMS_3(X) X(8, 2)
#define MPPI_MLP_DIMS_4(X) X(12, 6)
#define MPPI_MLP_DIMS_5(X) X(16, 8)
#if MPPI_MLP_GROUP == 0
#define MPPI_DIMS MPPI_MLP_DIMS_0
#define MPPI_THIS rollout_mlp_valu
#define MPPI_NEXT rollout_mlp_valu_g1
bool supported_mlp(int nx, int nu, int hidden) {
  if (hidden <= 0) return false;
#define X(NX, NU) if (nx == NX && nu == NU) return true;
  MPPI_MLP_DIMS_0(X) MPPI_MLP_DIMS_1(X) MPPI_MLP_DIMS_2(X) MPPI_MLP_DIMS_3(X) MPPI_MLP_DIMS_4(X) MPPI_MLP_DIMS_5(X)
#undef X
  return false;
}
#elif MPPI_MLP_GROUP == 1
#define MPPI_DIMS MPPI_MLP_DIMS_1
#define MPPI_THIS rollout_mlp_valu_g1
#define MPPI_NEXT rollout_mlp_valu_g2
#elif MPPI_MLP_GROUP == 2
#define MPPI_DIMS MPPI_MLP_DIMS_2
#define MPPI_THIS rollout_mlp_valu_g2
#define MPPI_NEXT rollout_mlp_valu_g3
#elif MPPI_MLP_GROUP == 3
#define MPPI_DIMS MPPI_MLP_DIMS_3
#define MPPI_THIS rollout_mlp_valu_g3
#define MPPI_NEXT rollout_mlp_valu_g4
#elif MPPI_MLP_GROUP == 4
#define MPPI_DIMS MPPI_MLP_DIMS_4
#define MPPI_THIS rollout_mlp_valu_g4
#define MPPI_NEXT rollout_mlp_valu_g5
#else
#define MPPI_DIMS MPPI_MLP_DIMS_5
#define MPPI_THIS rollout_mlp_valu_g5
#endif
#ifdef MPPI_NEXT
int MPPI_NEXT(const KArgs<float>& a, hipStream_t st);
int MPPI_NEXT(const KArgs<double>& a, hipStream_t st);
#endif
template <typename T> static int go(const KArgs<T>& a, hipStream_t st) {
  if (a.mp == nullptr || a.hidden <= 0) return MPPI_E_BADARG;
#define X(NX, NU) if (a.nx == NX && a.nu == NU) return launch_rollout<MlpModel<T, NX, NU>, T>(a, st);
  MPPI_DIMS(X)
#undef X
#ifdef MPPI_NEXT
  return MPPI_NEXT(a, st);
#else
  return MPPI_E_UNSUPPORTED;
#endif
}
#if MPPI_MLP_GROUP != 0
int MPPI_THIS(const KArgs<float>& a, hipStream_t st) { return go(a, st); }
int MPPI_THIS(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
}  // namespace mppi
#else
int rollout_mlp(const KArgs<float>& a, hipStream_t st) {
  if (a.W != nullptr) return MPPI_E_UNSUPPORTED;   // KMPPI: the matrix-core kernels read raw action rows (two-launch form)
  // fp32, hidden in {64,128,256}: matrix-core kernels.  Default: 16-bit MFMAs on split operands (bf16 x 3 for layer 1, fp16 x 2
  // for layer 2; fp32-level accuracy, rollout_mlp_split.hip: (nx,nu) = (16,4) and the further shapes of MPPI_SPLIT_DIMS_1); with
  // MPPI_MLP_EXACT=1 (or weights outside the fp16 operand range): the exact-fp32 kernel, bit-for-bit an fmaf chain -- the checker
  // of the former -- on the matrix cores for (16,4) (rollout_mlp_mfma.hip), per lane for the other shapes.  MPPI_MLP_VALU=1 forces
  // the per-lane form (A/B measurements).
  const char* fv = getenv("MPPI_MLP_VALU");
  const bool force_valu = fv != nullptr && fv[0] == '1';
  const char* fe = getenv("MPPI_MLP_EXACT");
  // ... or the host found weights outside the split kernel's fp16 operand range (MPPI_MODEL_FLAG_EXACT_FP32)
  const bool force_exact = (fe != nullptr && fe[0] == '1') || (a.model_flags & MPPI_MODEL_FLAG_EXACT_FP32) != 0;
  const bool smppi = a.B != nullptr || a.smooth_w != 0.f;      // lifted controls: the split kernel has the base sequence,
                                                               // the 1/dt rescaling and the smoothness cost; the exact one not
  if (!force_valu && a.M == 1 && a.states == nullptr) {
    const bool use_split = !force_exact && mlp_split_supported(a.nx, a.nu, a.hidden);
    const bool exact_ok = !smppi && mlp_mfma_supported(a.nx, a.nu, a.hidden);
    if (use_split || exact_ok) {
      // the matrix-core kernels read the engine's own layout: ask the caller to convert a (K,T,nu) draw
      if (a.noise_src == MPPI_NOISE_KTN) return MPPI_E_UNSUPPORTED;
      if (use_split) {
        const int r = rollout_mlp_split(a, st);
        if (r != MPPI_E_UNSUPPORTED) return r;
        // (nu != 4 with rows the split kernel does not read -- generated inside K1, a full Sigma coloured in the lane)
      }
      if (exact_ok) return rollout_mlp_mfma(a, st);
    }
  }
  return go(a, st);
}
int rollout_mlp(const KArgs<double>& a, hipStream_t st) { return go(a, st); }
}  // namespace mppi
#endif
