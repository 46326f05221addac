// Dev microbenchmark (never shipped): cycles per row-of-4 of the in-kernel normal generator, split
// into its parts.  hipcc --offload-arch=gfx950 -O3 -I../../include -I../../pytorch_mppi_amd/csrc rng_micro.hip -o rng_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.hpp"
using namespace mppi;

template <int MODE, int ROUNDS>
__global__ void __launch_bounds__(256) k(float* out, int rows, unsigned long long seed) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  unsigned xacc = 0;
  for (int jb = 0; jb < rows; ++jb) {
    U4 c{(unsigned)k, (unsigned)jb, 3u, 0u};
    U4 r;
    if constexpr (MODE == 1) {           // Box-Muller only (cheap counter hash instead of Philox)
      r = U4{c.x * 2654435761u + jb, c.y * 40503u + k, c.x ^ (jb * 97u), c.y + 77u * k};
    } else {
      unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
      constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
      for (int i = 0; i < ROUNDS; ++i) {
        unsigned long long p0 = (unsigned long long)M0 * c.x, p1 = (unsigned long long)M1 * c.z;
        U4 n{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n; k0 += W0; k1 += W1;
      }
      r = c;
    }
    if constexpr (MODE == 0) {           // Philox only
      xacc ^= r.x ^ r.y ^ r.z ^ r.w;
    } else {
      float a, b, d, e;
      box_muller(r.x, r.y, a, b);
      box_muller(r.z, r.w, d, e);
      acc += a + b + d + e;
    }
  }
  out[k] = acc + (float)xacc;
}

template <int MODE, int ROUNDS>
void run(const char* name, float* d, int rows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int K = 65536 * 4;      // 4 waves per SIMD: throughput, not latency
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, ROUNDS>), dim3(K / 256), dim3(256), 0, 0, d, rows, 1234ull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_rows = (double)K / 64 * rows;            // wave-level rows-of-4
  const double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_rows;   // SIMD-cycles per wave row at 2.4 GHz
  printf("%-28s %8.1f us  %6.1f cycles per wave-row-of-4   (%.2f T normals/s)\n", name, ms * 1e3, cyc,
         (double)K * rows * 4 / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 65536 * 4 * 4);
  const int rows = 192;
  run<0, 10>("philox4x32-10 only", d, rows);
  run<0, 7>("philox4x32-7 only", d, rows);
  run<1, 0>("box-muller only", d, rows);
  run<2, 10>("philox-10 + box-muller", d, rows);
  run<2, 7>("philox-7 + box-muller", d, rows);
  return 0;
}
