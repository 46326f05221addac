// loader_stream.hip -- how fast can ONE workgroup per CU pull the sample-minor rows-of-4 (the TNK4 array K1 streams)
// when a dedicated LOADER wave moves them global -> LDS with global_load_lds_dwordx4 and four consumer waves only read LDS?
// K1 (csrc/rollout.hpp) sits at the rate a CU pulls through global_load_dwordx4 into registers (~9.2 B/cycle/CU = 5.65 TB/s
// chip-wide); the guide quotes 6.4-6.8 TB/s for an LDS-DMA stream fed by one loader wave per CU.  Round 2
// tried LDS-DMA with every wave issuing its own rows (slower: the issue cost sits in line with the arithmetic).
// Geometry as K1 at C3: 256 samples per workgroup, 192 rows of 16 B per sample, rows HBM-cold (8 arrays cycled).
//   hipcc --offload-arch=gfx950 -O3 -I../../include -I../../pytorch_mppi_amd/csrc loader_stream.hip -o loader_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROWS = 192;

__device__ __forceinline__ void dma_row16(const float* row_base, unsigned lane_off_bytes, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off_bytes), "s"(row_base), "s"(lds_byte_addr)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// baseline: every lane loads its own rows straight into registers, DEPTH rows in flight (K1's register ring, simplified)
template <int DEPTH>
__global__ void __launch_bounds__(256) direct_kernel(const float* __restrict__ z, long long zp, float* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  float4 ring[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ring[d] = *reinterpret_cast<const float4*>(z + ((long long)d * zp + k) * 4);
  float acc = 0.f;
  for (int r0 = 0; r0 < ROWS; r0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const float4 v = ring[d];
      acc += (v.x + v.y) + (v.z + v.w);
      int rn = r0 + d + DEPTH;
      rn = rn < ROWS ? rn : ROWS - 1;
      asm volatile("" : "+v"(rn) : "v"(acc));          // refill behind the use
      ring[d] = *reinterpret_cast<const float4*>(z + ((long long)rn * zp + k) * 4);
    }
  }
  out[k] = acc;
}

// loader-fed: wave 4 streams the workgroup's rows into an LDS ring, waves 0-3 consume.
//   ready : rows [0, ready) have landed in LDS          (written by the loader, read by the consumers)
//   done[w]: consumer wave w has read rows [0, done[w])  (written by the consumers, read by the loader)
template <int RING, int AHEAD /* rows in flight, 4*AHEAD <= 63 */, int BATCH /* rows a consumer takes per flag check */, int MODE = 0>
__global__ void __launch_bounds__(320) loader_kernel(const float* __restrict__ z, long long zp, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(smem);      // [0] ready, [1..4] done
  float* ring = reinterpret_cast<float*>(smem + 64);                          // [RING][4 waves][64 lanes][4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 8) flags[threadIdx.x] = 0u;
  __syncthreads();
  const int k0 = blockIdx.x * 256;
  if (wave == 4) {
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)ring);
    for (int r = 0; r < ROWS; ++r) {
      if (MODE == 0 && r >= RING) {
        const unsigned need = (unsigned)(r - RING + 1);
        for (;;) {
          const unsigned d0 = flags[1], d1 = flags[2], d2 = flags[3], d3 = flags[4];
          const unsigned m01 = d0 < d1 ? d0 : d1, m23 = d2 < d3 ? d2 : d3;
          if ((m01 < m23 ? m01 : m23) >= need) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      const float* row = z + ((long long)r * zp + k0) * 4;
      const unsigned slot = (unsigned)(r % RING);
#pragma unroll
      for (int w = 0; w < 4; ++w) dma_row16(row + w * 256, (unsigned)lane * 16u, lds0 + (slot * 4u + (unsigned)w) * 1024u);
      if (r + 1 >= AHEAD) {
        wait_vmcnt<4 * (AHEAD - 1)>();                   // rows <= r - AHEAD + 1 have landed
        if (lane == 0) flags[0] = (unsigned)(r + 2 - AHEAD);
      }
    }
    wait_vmcnt<0>();
    if (lane == 0) flags[0] = (unsigned)ROWS;
  } else {
    float acc = 0.f;
    unsigned known = 0;
    if (MODE == 1) { out[k0 + wave * 64 + lane] = 0.f; return; }
    const float* mine = ring + wave * 256 + lane * 4;
    for (int r0 = 0; r0 < ROWS; r0 += BATCH) {
      const unsigned want = (unsigned)(r0 + BATCH < ROWS ? r0 + BATCH : ROWS);
      while (known < want) {
        known = flags[0];
        if (known < want) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const int r = r0 + b;
        if (r < ROWS) {
          const float4 v = *reinterpret_cast<const float4*>(mine + (r % RING) * 1024);
          acc += (v.x + v.y) + (v.z + v.w);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are in registers: the slots may be overwritten
      if (lane == 0) flags[1 + wave] = want;
    }
    out[k0 + wave * 64 + lane] = acc;
  }
}

template <class F>
double time_us(F&& launch, int n) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) launch(i);
  (void)hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch(i);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3;
}

int main() {
  const int K = 65536, NBUF = 8;
  const long long zp = K;
  const size_t elems = (size_t)ROWS * zp * 4;
  std::vector<float*> bufs(NBUF);
  std::vector<float> h(elems);
  for (size_t i = 0; i < elems; ++i) h[i] = (float)((i * 2654435761u >> 22) & 255) * (1.0f / 256.0f);
  for (int b = 0; b < NBUF; ++b) { (void)hipMalloc(&bufs[b], elems * 4); (void)hipMemcpy(bufs[b], h.data(), elems * 4, hipMemcpyHostToDevice); }
  float *o0, *o1; (void)hipMalloc(&o0, K * 4); (void)hipMalloc(&o1, K * 4);
  const double mb = elems * 4 / 1e6;
  auto report = [&](const char* name, double us, float* o) {
    std::vector<float> ho(K), ref(K);
    (void)hipMemcpy(ho.data(), o, K * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ref.data(), o0, K * 4, hipMemcpyDeviceToHost);
    double md = 0; for (int i = 0; i < K; ++i) md = fmax(md, fabs(ho[i] - ref[i]));
    hipError_t e = hipGetLastError();
    printf("%-78s %6.1f us  %5.2f TB/s  (%.1f %% of 8 TB/s)  max diff vs direct %g %s\n", name, us, mb / us, mb / us / 8.0 * 100, md,
           e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
  };
  double us = time_us([&](int i) { hipLaunchKernelGGL((direct_kernel<9>), dim3(K / 256), dim3(256), 0, 0, bufs[i % NBUF], zp, o0); }, 64);
  report("direct: own loads into a 9-row register ring (K1's form)", us, o0);
  us = time_us([&](int i) { hipLaunchKernelGGL((direct_kernel<16>), dim3(K / 256), dim3(256), 0, 0, bufs[i % NBUF], zp, o0); }, 64);
  report("direct: 16-row register ring (the reference for the checks below)", us, o0);
#define RUN(RING, AHEAD, BATCH) RUNM(RING, AHEAD, BATCH, 0)
#define RUNM(RING, AHEAD, BATCH, MODE)                                                                                     \
  {                                                                                                                 \
    const size_t smem = 64 + (size_t)RING * 4096;                                                                   \
    (void)hipFuncSetAttribute((const void*)loader_kernel<RING, AHEAD, BATCH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    us = time_us([&](int i) { hipLaunchKernelGGL((loader_kernel<RING, AHEAD, BATCH, MODE>), dim3(K / 256), dim3(320), smem, 0, bufs[i % NBUF], zp, o1); }, 64); \
    report("loader wave: ring " #RING " rows, " #AHEAD " in flight, consumers take " #BATCH " rows per check, mode " #MODE, us, o1); \
  }
  RUN(30, 15, 3)
  RUNM(30, 15, 3, 1)
  RUNM(30, 4, 3, 1)
  RUNM(30, 1, 3, 1)
  RUNM(30, 15, 3, 2)
  RUN(30, 2, 3)
  return 0;
}
