// k1ret_micro.hip -- prototype for VERDICT r02 item 4: a rng="philox" command that moves NO (K,T,nu) array.
// One launch: every lane generates the normals of ITS sample, rolls out, and keeps them ON CHIP until its weight
// is known -- AG_ROWS rows-of-4 in accumulation registers (one wave per SIMD: the other half of the unified
// 512-entry file is free anyway), LDS_ROWS rows in LDS ([row][thread][4]), the rest is generated a second
// time -- then forms its workgroup's partial record {beta_b, eta_b, P_b[j]} (the single-launch command's
// algebra, rollout.hpp FUSE block) from the retained rows.  Model: quad-toy integrator (nx 16, nu 12), C3.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I../../include -I../../pytorch_mppi_amd/csrc k1ret_micro.hip -o k1ret_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "common.hpp"
using namespace mppi;

constexpr int NX = 16, NU = 12, T_ = 64, J = T_ * NU, P4 = 3, ROWS = T_ * P4;
constexpr int TR = 16, NT = ROWS / TR;                   // tile of the weighting phase: 16 rows = 64 columns

struct Args {
  int K;
  unsigned long long seed, call;
  const float *U, *x0;
  float lambda_, sd, sinv, lo, hi;
  float *cost, *beta_part, *eta_part, *P_part;   // [K], [nb], [nb], [nb][J]
};

__device__ __forceinline__ float to_agpr(float v) {
  float r;
  asm volatile("; keep -> %0" : "=a"(r) : "0"(v));
  return r;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// Transposing wave reduction without the LDS crossbar: afterwards lane l holds the sum over the 64 lanes of v[l].
// s = 32 / 16: v_permlane32_swap / v_permlane16_swap exchange the halves of a register PAIR in one instruction
// (keep/send selects disappear); s = 8 .. 1: pair sums through DPP (row_ror:8, row_half_mirror, quad_perm) and one select.
__device__ __forceinline__ float wave_reduce_transpose64_dpp(float (&v)[64]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 32]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float X = v[i] + dpp_f<0x128>(v[i]), Y = v[i + 8] + dpp_f<0x128>(v[i + 8]);     // row_ror:8 == lane ^ 8
      v[i] = up ? Y : X;
    }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float X = v[i] + dpp_f<0x141>(v[i]), Y = v[i + 4] + dpp_f<0x141>(v[i + 4]);     // row_half_mirror: lane -> 7 - lane (crosses bit 2)
      v[i] = up ? Y : X;
    }
  }
  {
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float X = v[i] + dpp_f<0x4E>(v[i]), Y = v[i + 2] + dpp_f<0x4E>(v[i + 2]);       // quad_perm [2,3,0,1]
      v[i] = up ? Y : X;
    }
  }
  {
    const bool up = (lane & 1) != 0;
    const float X = v[0] + dpp_f<0xB1>(v[0]), Y = v[1] + dpp_f<0xB1>(v[1]);                 // quad_perm [1,0,3,2]
    v[0] = up ? Y : X;
  }
  return v[0];
}

// MODE 0: whole thing | 1: generate + rollout only (no retention, no weighting phase) | 2: generate only
// CS: timesteps per chunk of the generate + rollout loop (CS * 3 rows generated together)
template <int AG_ROWS, int LDS_ROWS, int MODE, int CS = 4, int RV = 0>
__global__ void __launch_bounds__(256) k1ret(const Args a) {
  constexpr int CR = CS * P4, NCH = T_ / CS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Ue = reinterpret_cast<float*>(smem);
  float* G = Ue + J;
  float* red = G + J;                 // [4]
  float* wsum = red + 4;              // [4][64]
  float4* keepL = reinterpret_cast<float4*>(wsum + 256);   // [LDS_ROWS][256]
  for (int j = threadIdx.x; j < J; j += 256) {
    const float u = a.U[j];
    Ue[j] = u;
    G[j] = a.lambda_ * u * a.sinv;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long seed = a.seed, call = a.call;
  float x[NX], rollout = 0.f, pert = 0.f;
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = a.x0[i];
  float keepA[AG_ROWS > 0 ? AG_ROWS * 4 : 4];
  __syncthreads();

  // ---- generate + roll out, chunk by chunk ----
  constexpr int NCA = (AG_ROWS + CR - 1) / CR;    // chunks that touch the accumulation registers (static indices)
  float zn[CR][4];
  if constexpr ((RV & 2) != 0) {
#pragma unroll
    for (int i = 0; i < CR; ++i) philox_normal4<float>(seed, call, k, (long long)i, zn[i]);
  }
  for (int c = 0; c < NCH; ++c) {
    float z[CR][4];
    if constexpr ((RV & 2) != 0) {
#pragma unroll
      for (int i = 0; i < CR; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) z[i][q] = zn[i][q];
      // the next chunk's rows: independent of this chunk's rollout -> the two instruction streams interleave
      // (rows past the horizon: generated, never used)
#pragma unroll
      for (int i = 0; i < CR; ++i) philox_normal4<float>(seed, call, k, (long long)(c + 1) * CR + i, zn[i]);
    } else {
#pragma unroll
      for (int i = 0; i < CR; ++i) philox_normal4<float>(seed, call, k, (long long)c * CR + i, z[i]);
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < CR; ++i) rollout += z[i][0] + z[i][1] + z[i][2] + z[i][3];
      continue;
    }
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      const int t = c * CS + s;
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        const float zz = z[s * P4 + n / 4][n % 4];
        float v = fmaf(zz, a.sd, Ue[t * NU + n]);
        v = clampT(v, a.lo, a.hi);
        const float e = v - Ue[t * NU + n];
        pert = fmaf(G[t * NU + n], e, pert);
        x[n] += v;
      }
      float cst = 0.f;
#pragma unroll
      for (int i = 0; i < NX; ++i) cst = fmaf(x[i], x[i], cst);
      rollout += cst;
    }
    if constexpr (MODE == 0 || MODE >= 3) {
      // stash the rows: accumulation registers (static row index -> a switch over the chunk), then LDS
      auto stash_dyn = [&](int i) {
        const int r = c * CR + i;
        if (r >= AG_ROWS && r < AG_ROWS + LDS_ROWS) keepL[(r - AG_ROWS) * 256 + threadIdx.x] = make_float4(z[i][0], z[i][1], z[i][2], z[i][3]);
      };
      bool done = false;
      static_for<0, NCA>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        if (c == C) {
          static_for<0, CR>([&](auto ii) {
            constexpr int I = decltype(ii)::value, R = C * CR + I;
            if constexpr (R < AG_ROWS) {
#pragma unroll
              for (int q = 0; q < 4; ++q) keepA[R * 4 + q] = R < 64 ? to_agpr(z[I][q]) : z[I][q];
            } else {
              stash_dyn(I);
            }
          });
          done = true;
        }
      });
      if (!done) {
#pragma unroll
        for (int i = 0; i < CR; ++i) stash_dyn(i);
      }
    }
  }
  const float total = rollout + pert;
  a.cost[k] = total;
  if constexpr (MODE == 1 || MODE == 2) return;
  if constexpr (MODE == 3) {   // timing only: generate + roll out + stash, keep the stash alive
    float q = 0.f;
    if (total == 123.456f) {
      static_for<0, (AG_ROWS > 0 ? AG_ROWS * 4 : 4)>([&](auto ii) { q += keepA[decltype(ii)::value]; });
      for (int i = 0; i < LDS_ROWS; ++i) q += keepL[i * 256 + threadIdx.x].x;
      a.cost[k] = q;
    }
    return;
  }

  // ---- the workgroup's part of the weighting (mppi.py:254-259, :268) relative to its own minimum ----
  const float inv_lambda = 1.f / a.lambda_;
  const float beta_b = block_min<float>(total, red);
  const float wk = __expf(-(total - beta_b) * inv_lambda);
  const float eta_b = block_sum<float>(wk, red);
  float psum[NT];
  for (int tile = 0; tile < NT; ++tile) {
    float zz[TR][4];
    bool got = false;
    static_for<0, (AG_ROWS + TR - 1) / TR>([&](auto tt) {
      constexpr int TI = decltype(tt)::value;
      static_assert(AG_ROWS % TR == 0, "whole tiles in the accumulation registers");
      if (tile == TI) {
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) zz[i][q] = keepA[(TI * TR + i) * 4 + q];
        got = true;
      }
    });
    if (!got) {
#pragma unroll
      for (int g = 0; g < TR / 4; ++g) {          // four rows at a time: LDS reads, or four interleaved Philox chains
        const int r0 = tile * TR + 4 * g;
        if (r0 + 3 < AG_ROWS + LDS_ROWS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 v = keepL[(r0 + i - AG_ROWS) * 256 + threadIdx.x];
            zz[4 * g + i][0] = v.x; zz[4 * g + i][1] = v.y; zz[4 * g + i][2] = v.z; zz[4 * g + i][3] = v.w;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (MODE == 4) { zz[4 * g + i][0] = zz[4 * g + i][1] = zz[4 * g + i][2] = zz[4 * g + i][3] = 0.f; }   // timing only
            else philox_normal4<float>(seed, call, k, (long long)r0 + i, zz[4 * g + i]);
          }
        }
      }
    }
    float acc[64];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = 4 * i + q;
        const float u = Ue[tile * 64 + cc];
        float v = fmaf(zz[i][q], a.sd, u);
        v = clampT(v, a.lo, a.hi);
        acc[cc] = wk * (v - u);
      }
    if constexpr ((RV & 1) == 0) {
      const float s = wave_reduce_transpose64<float>(acc);
      __syncthreads();
      wsum[wv * 64 + lane] = s;
      __syncthreads();
      if (threadIdx.x < 64) {
        const float sum = wsum[threadIdx.x] + wsum[64 + threadIdx.x] + wsum[128 + threadIdx.x] + wsum[192 + threadIdx.x];
        a.P_part[(long long)blockIdx.x * J + tile * 64 + threadIdx.x] = sum;
      }
    } else {
      psum[tile] = wave_reduce_transpose64_dpp(acc);
    }
  }
  if constexpr ((RV & 1) != 0) {
    // one combine over the four waves at the end: the kept rows are dead, their LDS serves as the exchange buffer
    float* ex = reinterpret_cast<float*>(keepL);          // [4][J]
    __syncthreads();
    static_for<0, NT>([&](auto tt) { ex[wv * J + decltype(tt)::value * 64 + lane] = psum[decltype(tt)::value]; });
    __syncthreads();
    for (int j = threadIdx.x; j < J; j += 256)
      a.P_part[(long long)blockIdx.x * J + j] = (ex[j] + ex[J + j]) + (ex[2 * J + j] + ex[3 * J + j]);
  }
  if (threadIdx.x == 0) {
    a.eta_part[blockIdx.x] = eta_b;
    a.beta_part[blockIdx.x] = beta_b;
  }
}

// checker: the same partial record, the obvious way (one thread per column, samples in order, rows regenerated)
__global__ void __launch_bounds__(256) ref_partial(const Args a, const float* cost, float* beta_part, float* eta_part, float* P_part) {
  __shared__ float red[4];
  __shared__ float w[256];
  const int k = blockIdx.x * 256 + threadIdx.x;
  const float total = cost[k];
  const float beta_b = block_min<float>(total, red);
  const float wk = __expf(-(total - beta_b) * (1.f / a.lambda_));
  w[threadIdx.x] = wk;
  const float eta_b = block_sum<float>(wk, red);
  __syncthreads();
  for (int j = threadIdx.x; j < J; j += 256) {
    float sum = 0.f;
    const float u = a.U[j];
    for (int s = 0; s < 256; ++s) {
      float r[4];
      philox_normal4<float>(a.seed, a.call, blockIdx.x * 256 + s, j / 4, r);
      float v = fmaf(r[j % 4], a.sd, u);
      v = clampT(v, a.lo, a.hi);
      sum += w[s] * (v - u);
    }
    P_part[(long long)blockIdx.x * J + j] = sum;
  }
  if (threadIdx.x == 0) { eta_part[blockIdx.x] = eta_b; beta_part[blockIdx.x] = beta_b; }
}

static std::vector<float> h_cost0;
static std::vector<float> h_Pref, h_eref, h_bref;

template <int AG_ROWS, int LDS_ROWS, int MODE, int CS = 4, int RV = 0>
void run(const char* name, const Args& a) {
  const int nb = a.K / 256;
  const size_t smem = (size_t)(2 * J + 4 + 256) * 4 + (size_t)(LDS_ROWS > 3 ? LDS_ROWS : 3) * 256 * 16;
  (void)hipFuncSetAttribute((const void*)k1ret<AG_ROWS, LDS_ROWS, MODE, CS, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n = 20;
  (void)hipMemset(a.P_part, 0, (size_t)nb * J * 4);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k1ret<AG_ROWS, LDS_ROWS, MODE, CS, RV>), dim3(nb), dim3(256), smem, 0, a);
  (void)hipEventRecord(e0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k1ret<AG_ROWS, LDS_ROWS, MODE, CS, RV>), dim3(nb), dim3(256), smem, 0, a);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  std::vector<float> c(a.K), P((size_t)nb * J), eta(nb), beta(nb);
  (void)hipMemcpy(c.data(), a.cost, a.K * 4, hipMemcpyDeviceToHost);
  double cs = 0; for (float v : c) cs += v;
  char verdict[256] = "";
  if (MODE == 2) {
    snprintf(verdict, sizeof verdict, "(generation only)");
  } else {
    if (h_cost0.empty()) h_cost0 = c;
    double md = 0; for (int i = 0; i < a.K; ++i) md = fmax(md, fabs(c[i] - h_cost0[i]));
    if (MODE == 0 || MODE == 4) {
      (void)hipMemcpy(P.data(), a.P_part, P.size() * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(eta.data(), a.eta_part, nb * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(beta.data(), a.beta_part, nb * 4, hipMemcpyDeviceToHost);
      double pe = 0, ps = 0, ee = 0, be = 0;
      for (size_t i = 0; i < P.size(); ++i) { pe = fmax(pe, fabs(P[i] - h_Pref[i])); ps = fmax(ps, fabs(h_Pref[i])); }
      for (int i = 0; i < nb; ++i) { ee = fmax(ee, fabs(eta[i] - h_eref[i]) / h_eref[i]); be = fmax(be, fabs(beta[i] - h_bref[i])); }
      snprintf(verdict, sizeof verdict, "cost diff %g | P err %.3g of max %.3g | eta rel %.3g | beta diff %g", md, pe, ps, ee, be);
    } else {
      snprintf(verdict, sizeof verdict, "cost diff %g", md);
    }
  }
  printf("%-58s %7.1f us   lds %6zu B  checksum %.6e  %s %s\n", name, ms / n * 1e3, smem, cs, verdict, e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main() {
  Args a;
  a.K = 65536; a.seed = 1234; a.call = 7; a.lambda_ = 40.f; a.sd = 1.f; a.sinv = 1.f; a.lo = -2.5f; a.hi = 2.5f;
  const int nb = a.K / 256;
  float *U, *x0;
  (void)hipMalloc(&U, J * 4); (void)hipMalloc(&x0, NX * 4);
  std::vector<float> hU(J), hx(NX);
  for (int j = 0; j < J; ++j) hU[j] = 0.02f * (float)((j * 37) % 11 - 5);
  for (int i = 0; i < NX; ++i) hx[i] = 0.1f * (float)(i - 8);
  (void)hipMemcpy(U, hU.data(), J * 4, hipMemcpyHostToDevice); (void)hipMemcpy(x0, hx.data(), NX * 4, hipMemcpyHostToDevice);
  a.U = U; a.x0 = x0;
  (void)hipMalloc(&a.cost, a.K * 4); (void)hipMalloc(&a.beta_part, nb * 4); (void)hipMalloc(&a.eta_part, nb * 4);
  (void)hipMalloc(&a.P_part, (size_t)nb * J * 4);
  run<0, 0, 2>("generation only (12 rows per block, 1 wave/SIMD)", a);
  run<0, 0, 1>("generate + roll out (no retention, no weighting)", a);
  // reference partial records from the costs just written
  {
    float *bp, *ep, *pp;
    (void)hipMalloc(&bp, nb * 4); (void)hipMalloc(&ep, nb * 4); (void)hipMalloc(&pp, (size_t)nb * J * 4);
    hipLaunchKernelGGL(ref_partial, dim3(nb), dim3(256), 0, 0, a, a.cost, bp, ep, pp);
    h_Pref.resize((size_t)nb * J); h_eref.resize(nb); h_bref.resize(nb);
    (void)hipMemcpy(h_Pref.data(), pp, h_Pref.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h_eref.data(), ep, nb * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h_bref.data(), bp, nb * 4, hipMemcpyDeviceToHost);
  }
  run<0, 0, 0, 4, 1>("nothing kept; DPP reduce, one combine", a);
  run<64, 36, 0, 4, 1>("64 acc + 36 LDS; DPP reduce, one combine", a);
  run<0, 0, 1, 2, 2>("PIPELINED generate + roll out, chunks of 2", a);
  run<0, 0, 1, 4, 2>("PIPELINED generate + roll out, chunks of 4", a);
  run<0, 0, 1, 1, 2>("PIPELINED generate + roll out, chunks of 1", a);
  run<64, 36, 0, 2, 3>("PIPELINED 64 acc + 36 LDS, chunks of 2; DPP reduce, one combine", a);
  run<64, 36, 0, 4, 3>("PIPELINED 64 acc + 36 LDS, chunks of 4; DPP reduce, one combine", a);
  run<64, 36, 3, 4, 1>("TIMING: 64 acc + 36 LDS, generate + roll out + stash only", a);
  run<64, 0, 3, 4, 1>("TIMING: 64 acc, generate + roll out + stash only", a);
  run<0, 36, 3, 4, 1>("TIMING: 36 LDS, generate + roll out + stash only", a);
  run<0, 36, 4, 4, 1>("TIMING: 36 LDS, weighting without the second generation (P wrong)", a);
  run<64, 36, 4, 4, 1>("TIMING: 64 acc + 36 LDS, weighting without the second generation (P wrong)", a);
  run<0, 0, 4, 4, 1>("TIMING: nothing kept, weighting without the second generation (P wrong)", a);
  return 0;
}
