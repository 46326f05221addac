// k1gen_micro.hip -- prototype for VERDICT r02 item 4 (break the 3x traffic of a rng="philox" command):
// K1 that GENERATES its normals at (close to) the chip's RNG rate instead of reading rows a generator launch
// wrote.  At C3 (K = 65536) lane = sample gives ONE wave per SIMD, and one wave cannot issue Philox + Box-Muller
// + the rollout fast enough (the product's philox-k1 form: 60-67 us against 34.7 + 33.6 us for generator + K1).
// Here a workgroup is 512 threads for 256 samples: waves 0-3 ("A") roll out, waves 4-7 ("B", same SIMDs) only
// generate; B owns BSHARE of every 3 rows-of-4 and hands them to its A through an LDS double buffer, A
// generates the rest itself; every generated row is also stored to the TNK4 array for K3.  Two waves per SIMD,
// both busy: the target is RNG pipe time (~33 us) + rollout issue (~8 us).
// Model: the quad-toy integrator (nx 16, nu 12, cost sum x^2), diagonal Sigma, T = 64 -- config C3.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I../../include -I../../pytorch_mppi_amd/csrc k1gen_micro.hip -o k1gen_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.hpp"
using namespace mppi;

constexpr int NX = 16, NU = 12, T_ = 64, J = T_ * NU, P4 = 3;   // 3 rows-of-4 per timestep

struct Args {
  int K;
  long long zp;
  unsigned long long seed, call;
  const float *U, *x0;
  float lambda_, sd, sinv, lo, hi;
  float *z, *cost;
};

__device__ __forceinline__ void step(const Args& a, const float* Ue, const float* G, int t, const float (&z)[NU], float (&x)[NX],
                                     float& rollout, float& pert) {
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    float v = fmaf(z[n], a.sd, Ue[t * NU + n]);
    v = clampT(v, a.lo, a.hi);
    const float e = v - Ue[t * NU + n];
    pert = fmaf(G[t * NU + n], e, pert);
    x[n] += v;
  }
  float c = 0.f;
#pragma unroll
  for (int i = 0; i < NX; ++i) c = fmaf(x[i], x[i], c);
  rollout += c;
}

// BSHARE = 0: the one-role form (256 threads, every lane generates all its rows) = today's structure
template <int BSHARE, int NSS>
__global__ void __launch_bounds__(BSHARE == 0 ? 256 : 512) k1gen(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ue = reinterpret_cast<float*>(smem);
  float* G = Ue + J;
  float* ring = G + J;                       // [2][4 pairs][NB rows][64 lanes][4]
  constexpr int R = NSS * P4;                // rows of a block
  constexpr int NB = R * BSHARE / 3;         // of which B generates
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    const float u = a.U[j];
    Ue[j] = u;
    G[j] = a.lambda_ * u * a.sinv;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pair = wave & 3, role = BSHARE == 0 ? 0 : wave >> 2;
  const int k = blockIdx.x * 256 + pair * 64 + lane;
  auto slot = [=](int buf, int i) { return ring + ((((buf * 4 + pair) * (NB > 0 ? NB : 1)) + i) * 64 + lane) * 4; };
  auto owned_by_b = [](int r) { return (r % 3) < BSHARE; };
  auto b_index = [](int r) { return (r / 3) * BSHARE + ((r % 3) < BSHARE ? (r % 3) : BSHARE); };   // B-owned rows in front of row r
  const unsigned long long seed = a.seed, call = a.call;
  float* const zout = a.z;
  const long long zp = a.zp;
  auto gen_row = [=](long long jb, float (&r)[4]) {
    philox_normal4<float>(seed, call, k, jb, r);
    *reinterpret_cast<float4*>(zout + (jb * zp + k) * 4) = make_float4(r[0], r[1], r[2], r[3]);
  };
  auto produce = [=](int blk, int buf) {     // B: its rows of block blk -> LDS (+ global)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (owned_by_b(r)) {
        float v[4];
        gen_row((long long)blk * R + r, v);
        *reinterpret_cast<float4*>(slot(buf, b_index(r))) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  constexpr int NBLK = T_ / NSS;
  if (role == 1) produce(0, 0);
  __syncthreads();                           // tables + block 0
  float x[NX], rollout = 0.f, pert = 0.f;
  if (role == 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = a.x0[i];
  }
  for (int blk = 0; blk < NBLK; ++blk) {
    if (role == 1) {
      if (blk + 1 < NBLK) produce(blk + 1, (blk + 1) & 1);
    } else {
#pragma unroll
      for (int s = 0; s < NSS; ++s) {
        float z[NU];
#pragma unroll
        for (int q = 0; q < P4; ++q) {
          const int r = s * P4 + q;
          float v[4];
          if (owned_by_b(r)) {
            const float4 w = *reinterpret_cast<const float4*>(slot(blk & 1, b_index(r)));
            v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
          } else {
            gen_row((long long)blk * R + r, v);
          }
          z[4 * q] = v[0]; z[4 * q + 1] = v[1]; z[4 * q + 2] = v[2]; z[4 * q + 3] = v[3];
        }
        step(a, Ue, G, blk * NSS + s, z, x, rollout, pert);
      }
    }
    if (BSHARE != 0) __syncthreads();
  }
  if (role == 0) a.cost[k] = rollout + pert;
}

template <int BSHARE, int NSS>
double run(const char* name, const Args& a, std::vector<float>& cost_out) {
  constexpr int threads = BSHARE == 0 ? 256 : 512;
  constexpr int NB = NSS * P4 * BSHARE / 3;
  const size_t smem = (size_t)(2 * J + 2 * 4 * (NB > 0 ? NB : 1) * 64 * 4) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)k1gen<BSHARE, NSS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 20;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k1gen<BSHARE, NSS>), dim3(a.K / 256), dim3(threads), smem, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k1gen<BSHARE, NSS>), dim3(a.K / 256), dim3(threads), smem, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  cost_out.resize(a.K);
  hipMemcpy(cost_out.data(), a.cost, a.K * 4, hipMemcpyDeviceToHost);
  double s = 0; for (float c : cost_out) s += c;
  printf("%-44s %7.1f us per launch (back to back, incl. ~1.5 us boundary)   lds %6zu B   checksum %.6e  %s\n", name, ms / n * 1e3, smem, s,
         e == hipSuccess ? "" : hipGetErrorString(e));
  return ms / n * 1e3;
}

int main() {
  Args a;
  a.K = 65536; a.zp = a.K; a.seed = 1234; a.call = 7; a.lambda_ = 40.f; a.sd = 1.f; a.sinv = 1.f; a.lo = -1e30f; a.hi = 1e30f;
  float *U, *x0;
  hipMalloc(&U, J * 4); hipMalloc(&x0, NX * 4);
  std::vector<float> hU(J), hx(NX);
  for (int j = 0; j < J; ++j) hU[j] = 0.02f * (float)((j * 37) % 11 - 5);
  for (int i = 0; i < NX; ++i) hx[i] = 0.1f * (float)(i - 8);
  hipMemcpy(U, hU.data(), J * 4, hipMemcpyHostToDevice); hipMemcpy(x0, hx.data(), NX * 4, hipMemcpyHostToDevice);
  a.U = U; a.x0 = x0;
  hipMalloc(&a.z, (size_t)192 * a.K * 16); hipMalloc(&a.cost, a.K * 4);
  std::vector<float> c0, c1;
  run<0, 4>("one role (256 thr): every lane generates all", a, c0);
  run<2, 4>("A/B roles, B owns 2 of 3 rows, block 4 steps", a, c1);
  double md = 0; for (int i = 0; i < a.K; ++i) md = fmax(md, fabs(c0[i] - c1[i]));
  printf("   max |cost difference| vs one-role: %g\n", md);
  run<2, 8>("A/B roles, B owns 2 of 3 rows, block 8 steps", a, c1);
  run<3, 4>("A/B roles, B owns all rows,    block 4 steps", a, c1);
  md = 0; for (int i = 0; i < a.K; ++i) md = fmax(md, fabs(c0[i] - c1[i]));
  printf("   max |cost difference| vs one-role: %g\n", md);
  run<1, 4>("A/B roles, B owns 1 of 3 rows, block 4 steps", a, c1);
  run<3, 8>("A/B roles, B owns all rows,    block 8 steps", a, c1);
  return 0;
}
