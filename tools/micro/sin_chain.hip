// sin_chain.hip -- a rollout is one dependent chain per lane: what does a sinf in that chain cost, the library's against the branch-free one
// of csrc/common.hpp (m_sin), at one wave per SIMD and at eight?   hipcc --offload-arch=gfx950 -O3 -I include -I pytorch_mppi_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.hpp"
using namespace mppi;
template <int WHICH>
__global__ void chain(float* out, int n) {
  float x = 0.001f * (threadIdx.x + blockIdx.x * blockDim.x % 977), acc = 0.f;
  for (int i = 0; i < n; ++i) {
    const float s = WHICH == 0 ? sinf(x) : (WHICH == 1 ? m_sin_moderate(x) : m_sin(x));
    x = fmaf(s, 0.7f, x * 0.99f + 0.3f);
    acc += s;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + x;
}
int main() {
  float* d; (void)hipMalloc(&d, 4 << 20);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int n = 2000;
  for (int waves : {1, 8}) {
    const int blocks = 256 * waves;      // 256 threads = one wave per SIMD of a CU
    auto run = [&](auto kern, const char* name) {
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, n);
      (void)hipEventRecord(a, 0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, n);
      (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      printf("%d wave(s) per SIMD  %-34s %7.1f ns per chained call (incl. 3 other ops)\n", waves, name, ms / 5 * 1e6 / n);
    };
    run(chain<0>, "ocml sinf");
    run(chain<1>, "m_sin_moderate (ocml, |x| < 8192)");
    run(chain<2>, "m_sin (branch-free, this round)");
  }
  return 0;
}
