// event_clock.hip -- which host-visible clock equals rocprofv3's kernel duration?
// (bench.py's roofline must follow from the committed rocprofv3 trace to +-1 %.)
//
// Three kernels back to back on one stream: marker A, the kernel under test K (spins for a given
// number of wall_clock64 ticks in every workgroup, 256 workgroups), marker B.  Each launch carries a
// hipExtLaunchKernelGGL STOP event only (the event is bound to the kernel's own dispatch; a START event
// would be a separate marker packet in front of it).  hipEventElapsedTime(X, Y) of two kernel-bound
// events is Y.end - X.start, so
//     dur(K) = el(eA, eK) + el(eK, eB) - el(eA, eB) = K.end - K.start
// -- the dispatch's own begin/end timestamps, what rocprofv3 --kernel-trace reports.  The program prints
// that figure, the classic start/stop-event pair, and the kernel's own span on the device wall clock
// (min entry .. max exit), per repetition; run it under `rocprofv3 --kernel-trace --stats` and compare
// with the trace's duration of spin_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o event_clock event_clock.hip && ./event_clock [ticks] [reps]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void marker_kernel() {}

__global__ void __launch_bounds__(256) spin_kernel(unsigned long long ticks, unsigned long long* ts) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) atomicMin(&ts[0], t0);
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&ts[1], (unsigned long long)wall_clock64());
}

int main(int argc, char** argv) {
  const unsigned long long ticks = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3300;   // 100 MHz clock: 33 us
  const int reps = argc > 2 ? atoi(argv[2]) : 40;
  int khz = 100000;
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned long long* ts;
  CK(hipMalloc(&ts, 16 * reps));
  std::vector<unsigned long long> init(2 * reps);
  for (int i = 0; i < reps; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
  CK(hipMemcpy(ts, init.data(), 16 * reps, hipMemcpyHostToDevice));
  std::vector<hipEvent_t> eA(reps), eK(reps), eB(reps), s0(reps), s1(reps);
  for (int i = 0; i < reps; ++i) { CK(hipEventCreate(&eA[i])); CK(hipEventCreate(&eK[i])); CK(hipEventCreate(&eB[i])); CK(hipEventCreate(&s0[i])); CK(hipEventCreate(&s1[i])); }
  // warm-up
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, st, ticks, ts);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ts, init.data(), 16 * reps, hipMemcpyHostToDevice));
  for (int i = 0; i < reps; ++i) {
    if (i % 2 == 0) {
      // stop events only, bound to three consecutive dispatches
      hipExtLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, st, nullptr, eA[i], 0);
      hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, st, nullptr, eK[i], 0, ticks, ts + 2 * i);
      hipExtLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, st, nullptr, eB[i], 0);
    } else {
      // the classic pair attached to the launch
      hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, st, s0[i], s1[i], 0, ticks, ts + 2 * i);
    }
  }
  CK(hipStreamSynchronize(st));
  std::vector<unsigned long long> h(2 * reps);
  CK(hipMemcpy(h.data(), ts, 16 * reps, hipMemcpyDeviceToHost));
  std::vector<double> exact, pair, dev_e, dev_p;
  for (int i = 0; i < reps; ++i) {
    const double dev = (double)(h[2 * i + 1] - h[2 * i]) / khz * 1e3;   // us
    if (i % 2 == 0) {
      float ak = 0, kb = 0, ab = 0;
      CK(hipEventElapsedTime(&ak, eA[i], eK[i]));
      CK(hipEventElapsedTime(&kb, eK[i], eB[i]));
      CK(hipEventElapsedTime(&ab, eA[i], eB[i]));
      const double d = ((double)ak + kb - ab) * 1e3;
      if (i < 8) printf("rep %2d  triple: el(A,K) %.3f  el(K,B) %.3f  el(A,B) %.3f  -> dur(K) %.3f us | device span %.3f us\n", i, ak * 1e3, kb * 1e3, ab * 1e3, d, dev);
      exact.push_back(d); dev_e.push_back(dev);
    } else {
      float p = 0;
      CK(hipEventElapsedTime(&p, s0[i], s1[i]));
      if (i < 8) printf("rep %2d  pair  : el(start,stop) %.3f us | device span %.3f us\n", i, p * 1e3, dev);
      pair.push_back(p * 1e3); dev_p.push_back(dev);
    }
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto avg = [](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return s / v.size(); };
  printf("SUMMARY ticks=%llu (%.2f us)  triple-stop-event dur(K): median %.3f avg %.3f us | start/stop pair: median %.3f avg %.3f us | "
         "device span: median %.3f (triple launches) %.3f (pair launches) us\n",
         ticks, (double)ticks / khz * 1e3, med(exact), avg(exact), med(pair), avg(pair), med(dev_e), med(dev_p));
  printf("compare with rocprofv3 --kernel-trace: spin_kernel launches alternate triple (even) / pair (odd) after 3 warm-ups\n");
  return 0;
}
