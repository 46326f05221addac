// Standalone check (GPU) of the operand layout assumed by csrc/rollout_mlp_split.hip for
// v_mfma_f32_16x16x32_bf16:  A[i][k]: lane l holds i = l & 15, k = 8*(l >> 4) + e (e = 0..7, two per VGPR,
// low half first);  B[k][j]: lane l holds j = l & 15, k = 8*(l >> 4) + e;  D[i][j]: lane l holds
// j = l & 15, i = 4*(l >> 4) + r.  A = asymmetric random bf16-exact integers, B likewise; compares with
// a host triple loop.   hipcc --offload-arch=gfx950 -O2 mfma_bf16_layout.hip -o mfma_bf16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* A /*16x32*/, const float* B /*32x16*/, float* D /*16x16*/) {
  const int l = threadIdx.x, i = l & 15, kb = l >> 4;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (short)(__float_as_uint(A[i * 32 + 8 * kb + e]) >> 16);
    b[e] = (short)(__float_as_uint(B[(8 * kb + e) * 16 + i]) >> 16);
  }
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * kb + r) * 16 + i] = d[r];
}

int main() {
  float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 17 - 8); hB[i] = (float)(rand() % 13 - 6); }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0;
      for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 16 + j];
      ref[i * 16 + j] = s;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("mfma_f32_16x16x32_bf16 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
