// Dev microbenchmark (never shipped): does VALU work (v_exp + v_rcp) overlap with MFMAs issued by
// another wave of the same SIMD?  fp32 16x16x4 vs bf16 16x16x16, 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// MODE bit0: MFMA work, bit1: VALU (tanh-like) work.  KIND 0: fp32 16x16x4, 1: bf16 16x16x16
template <int MODE, int KIND>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.1f * i + lane * 1e-3f;
  s16x4 ab = {(short)(0x3f80 + lane), (short)0x3f00, (short)0x3e80, (short)0x3f80};
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, ab, acc[i], 0, 0, 0);
        }
      }
    }
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[i]) + 1.0f);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int KIND>
float run(float* d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, d, iters);   // 8 waves per CU = 2 per SIMD
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e3f;
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  const int iters = 4000;
  for (int kind = 0; kind < 2; ++kind) {
    float m, v, b;
    if (kind == 0) { m = run<1, 0>(d, iters); v = run<2, 0>(d, iters); b = run<3, 0>(d, iters); }
    else { m = run<1, 1>(d, iters); v = run<2, 1>(d, iters); b = run<3, 1>(d, iters); }
    // per iteration and wave: 16 MFMAs, 8 x (exp + add + rcp)
    const double cyc = 2.4e9 * 1e-6 / iters / 2;    // us -> SIMD cycles per iteration per wave (2 waves per SIMD share it)
    printf("%s: MFMA only %7.1f us (%.1f cyc/MFMA)  VALU only %7.1f us (%.1f cyc per exp+add+rcp)  both %7.1f us  (sum %.1f, max %.1f)\n",
           kind == 0 ? "fp32 16x16x4  " : "bf16 16x16x16 ", m, m * cyc / 16, v, v * cyc / 8, b, m + v, m > v ? m : v);
  }
  return 0;
}
