// Dev microbenchmark (never shipped): what the 32x32 -> 64-bit products of a Philox4x32-10 round cost on gfx950, by spelling.
//   A  v_mul_hi_u32 + v_mul_lo_u32 (what hipcc emits for (u64)M * x)         -- the product kernel's form
//   B  one v_mad_u64_u32 (full 64-bit product + 0) per product, inline asm
//   C  v_mul_hi_u32 + v_mul_lo_u32 with the 3-way XOR as v_bitop3_b32 (= csrc/common.hpp today)
//   D  B + bitop3
// at 4 waves per SIMD (throughput) and at 1 wave per SIMD (what the on-chip K1 runs at), Philox only and Philox + Box-Muller.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I pytorch_mppi_amd/csrc tools/micro/philox_mul_micro.hip -o tools/micro/philox_mul_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.hpp"
using namespace mppi;

template <bool MAD>
__device__ __forceinline__ void mulhilo(unsigned m, unsigned x, unsigned& hi, unsigned& lo) {
  if constexpr (MAD) {
    unsigned long long r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "s"(m), "v"(x) : "vcc");
    hi = (unsigned)(r >> 32);
    lo = (unsigned)r;
  } else {
    const unsigned long long p = (unsigned long long)m * x;
    hi = (unsigned)(p >> 32);
    lo = (unsigned)p;
  }
}

template <bool MAD, bool BITOP>
__device__ __forceinline__ U4 philox(U4 c, unsigned k0, unsigned k1) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned h0, l0, h1, l1;
    mulhilo<MAD>(M0, c.x, h0, l0);
    mulhilo<MAD>(M1, c.z, h1, l1);
    U4 n;
    n.x = BITOP ? __builtin_amdgcn_bitop3_b32(h1, c.y, k0, 0x96) : (h1 ^ c.y ^ k0);
    n.y = l1;
    n.z = BITOP ? __builtin_amdgcn_bitop3_b32(h0, c.w, k1, 0x96) : (h0 ^ c.w ^ k1);
    n.w = l0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

template <bool MAD, bool BITOP, bool BM>
__global__ void __launch_bounds__(256) k(float* out, int rows, unsigned long long seed, unsigned* check) {
  const int kk = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  unsigned xacc = 0;
  for (int jb = 0; jb < rows; jb += 4) {
    U4 r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = philox<MAD, BITOP>(U4{(unsigned)kk, (unsigned)(jb + q), 3u, 0u}, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (BM) {
        float a, b, d, e;
        box_muller(r[q].x, r[q].y, a, b);
        box_muller(r[q].z, r[q].w, d, e);
        acc += a + b + d + e;
      }
      xacc ^= r[q].x ^ r[q].y ^ r[q].z ^ r[q].w;
    }
  }
  out[kk] = acc;
  check[kk] = xacc;
}

template <bool MAD, bool BITOP, bool BM>
unsigned run(const char* name, float* d, unsigned* chk, int rows, int K) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MAD, BITOP, BM>), dim3(K / 256), dim3(256), 0, 0, d, rows, 1234ull, chk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  unsigned c0; (void)hipMemcpy(&c0, chk + 12345, 4, hipMemcpyDeviceToHost);
  const double wave_rows = (double)K / 64 * rows;
  printf("%-44s K %7d (%d wave/SIMD): %8.1f us  %6.1f cycles per wave row-of-4 @2.4GHz  check %08x\n", name, K, K / 65536, best * 1e3,
         best * 1e-3 * 2.4e9 * 1024 / wave_rows, c0);
  return c0;
}

int main() {
  float* d; unsigned* chk;
  (void)hipMalloc(&d, 65536 * 4 * 4); (void)hipMalloc(&chk, 65536 * 4 * 4);
  const int rows = 192;
  for (int K : {65536, 262144}) {
    const unsigned a = run<false, false, false>("A mul_hi + mul_lo, two xors, philox only", d, chk, rows, K);
    const unsigned b = run<true, false, false>("B v_mad_u64_u32, two xors, philox only", d, chk, rows, K);
    const unsigned c = run<false, true, false>("C mul_hi + mul_lo, bitop3, philox only", d, chk, rows, K);
    const unsigned e = run<true, true, false>("D v_mad_u64_u32, bitop3, philox only", d, chk, rows, K);
    if (a != b || a != c || a != e) printf("  !! streams differ\n");
    run<false, true, true>("C + box-muller (the engine's generator)", d, chk, rows, K);
    run<true, true, true>("D + box-muller", d, chk, rows, K);
  }
  return 0;
}
