// K1 for the 2-layer MLP residual model on the matrix cores (fp32, exact: v_mfma_f32_16x16x4_f32
// is an fmaf chain) -- BASELINE.json configs[3..4]: nx=16, nu=4, hidden=256.
//
//   x' = x + s * (W2 tanh(W1 [x;u] + b1) + b2),   cost = sum x^2
//
// Transposed chain, samples on the MFMA column axis (16 samples per tile, NT tiles per wave):
//   layer 1   H^T (H x 16)  = W1 (H x 20)  . [x;u]^T (20 x 16)     A = W1 tile, B = inputs
//   layer 2   O^T (16 x 16) = W2 (16 x H)  . tanh(H)^T (H x 16)    A = W2 tile, B = tanh(H)
// D layout of 16x16x4: lane (g = lane>>4, s = lane&15) holds rows 4g+r (r<4) of column s.
// B layout: lane (g, s) supplies k-index g of the current k-step for column s.
// So a layer-1 accumulator register r of hidden tile m (rows 16m+4g+r) IS the B operand of the
// layer-2 k-step "(m, r)" if W2's columns are fetched in that order, and the layer-2 output
// (state rows 4g+r) IS the B operand of the next timestep's layer-1 k-step "r" if W1's columns
// are fetched as 4g+r: no lane permutes, no LDS round trip, between layers or between steps.
// All weights stay in registers for the whole horizon: H/16*5 + H/4 VGPRs per lane (144 at H=256).
// Only r = 1/(2^x + 1) of tanh = 1 - 2r runs on the VALU (v_exp + v_add + v_rcp); the affine parts live in the weights.
#include <hip/hip_ext.h>
#include "actions.hpp"
#include "dispatch.hpp"

namespace mppi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// tanh(x) = 1 - 2 r(x),  r = 1 / (2^{x * 2 log2 e} + 1).  Every VALU cycle here is additive to the
// kernel time: the fp32 MFMA runs at the fp32 vector rate and (measured, tools/run_c4_variants.sh)
// does not overlap VALU work -- 597 us without tanh, 857 us with the five-instruction form at C4.
// So the affine parts are folded into the weights once per launch and only r is evaluated per
// hidden unit (v_exp + v_add + v_rcp):
//   layer 1 is loaded as  (2 log2 e) * [W1 | b1]          -> the pre-activation arrives in exp2 units
//   layer 2 is loaded as  -2 * W2,  b2 + rowsum(W2)        -> W2 tanh(h) + b2 = (-2 W2) r + (b2 + W2 1)
// No clamp needed: 2^x -> inf gives rcp(inf) = 0 (tanh = 1), 2^x -> 0 gives r = 1 (tanh = -1).
constexpr float MLP_EXP2_SCALE = 2.8853900817779268f;   // 2 * log2(e)
__device__ __forceinline__ float half_one_minus_tanh(float x_exp2_units) {
#ifdef MPPI_MLP_NOTANH   // experiment only (tools/): MFMA pipe alone
  return x_exp2_units;
#endif
  return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x_exp2_units) + 1.0f);
}

constexpr int MLP_NX = 16, MLP_NU = 4, MLP_NI = 20;
#ifndef MPPI_MLP_NT
#define MPPI_MLP_NT 2     // sample tiles (of 16) per wave; a workgroup always covers 256 samples
#endif
// NT = 2 -> 8 waves per workgroup = TWO waves per SIMD: one wave's tanh (VALU) runs while the other
// wave's MFMAs occupy the matrix pipe; each wave still has 2 independent accumulator chains.
constexpr int MLP_NT = MPPI_MLP_NT;
constexpr int MLP_THREADS = 256 / (16 * MLP_NT) * WAVE;

// component g (lane-dependent) of a row held in registers, as two levels of v_cndmask.  The operands
// are pinned in VGPRs: left alone, hipcc turns the select into a dynamically indexed load of the
// array, which forces the row through scratch memory every timestep.
__device__ __forceinline__ float pick4(const float (&z)[4], int g) {
  float a0 = z[0], a1 = z[1], a2 = z[2], a3 = z[3];
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  const float lo = (g & 1) ? a1 : a0, hi = (g & 1) ? a3 : a2;
  return (g & 2) ? hi : lo;
}

template <int HT /* hidden / 16 */, int NOISE, bool DIAG>
__global__ void __launch_bounds__(MLP_THREADS) rollout_mlp_mfma_kernel(const KArgs<float> a_in) {
  constexpr int NU = MLP_NU, NX = MLP_NX, NT = MLP_NT, H = HT * 16;
  const KArgs<float> a = env_view(a_in);
  stamp_entry(a.tstamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Ue = reinterpret_cast<float*>(smem_raw);   // [J]
  float* Um = Ue + a.J;                             // [J]
  float* G = Um + a.J;                              // [J]
  float* red = G + a.J;                             // [waves per workgroup]
  float* b1s = red + MLP_THREADS / WAVE;            // [H]
  float* fac = b1s + H;                             // [2*NU*NU]

  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int g = lane >> 4, s = lane & 15;

  // ---- parameters: blob = W1 (H,20) | b1 (H) | W2 (16,H) | b2 (16) | res_scale ----
  const float* __restrict__ W1 = a.mp;
  const float* __restrict__ b1 = W1 + H * MLP_NI;
  const float* __restrict__ W2 = b1 + H;
  const float* __restrict__ b2 = W2 + NX * H;
  const float rs = b2[NX];
  float qx4[4];                                      // cost weights (blob: ... | res_scale | qx (16) | qu (4)): rows 4g + r, control g
#pragma unroll
  for (int r = 0; r < 4; ++r) qx4[r] = b2[NX + 1 + 4 * g + r];
  const float qu_g = b2[NX + 1 + NX + g];
  float w1r[HT][5], w2r[HT][4], b2r[4];
#pragma unroll
  for (int m = 0; m < HT; ++m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w1r[m][q] = MLP_EXP2_SCALE * W1[(16 * m + s) * MLP_NI + 4 * g + q];
    w1r[m][4] = MLP_EXP2_SCALE * W1[(16 * m + s) * MLP_NI + NX + g];
#pragma unroll
    for (int r = 0; r < 4; ++r) w2r[m][r] = W2[s * H + 16 * m + 4 * g + r];
  }
  // rowsum(W2): this lane holds a quarter of row s; the three other quarters sit in lanes s + 16 g'
  float rowsum = 0.f;
#pragma unroll
  for (int m = 0; m < HT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) rowsum += w2r[m][r];
  }
  rowsum += __shfl_xor(rowsum, 16, WAVE);
  rowsum += __shfl_xor(rowsum, 32, WAVE);
#pragma unroll
  for (int r = 0; r < 4; ++r) b2r[r] = b2[4 * g + r] + __shfl(rowsum, 4 * g + r, WAVE);   // D rows are 4g+r
#pragma unroll
  for (int m = 0; m < HT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) w2r[m][r] *= -2.0f;
  }

  ActionConsts<float, NU> ac;
  ac.load(a, DIAG ? nullptr : fac);
  for (int j = threadIdx.x; j < a.J; j += MLP_THREADS) Ue[j] = u_eff(a, j);
  for (int h = threadIdx.x; h < H; h += MLP_THREADS) b1s[h] = MLP_EXP2_SCALE * b1[h];
  __syncthreads();
  for (int j = threadIdx.x; j < a.J; j += MLP_THREADS) {
    const int n = j % NU, t0 = j - n;
    const float uj = Ue[j];
    Um[j] = a.coloured ? uj : uj + a.mu[n];
    float gg;
    if constexpr (DIAG) {
      if (a.coloured && !a.diag) {          // generator-coloured full Sigma: whole-row G from global
        gg = 0.f;
        for (int m = 0; m < NU; ++m) gg = fmaf(a.sinv[n * NU + m], Ue[t0 + m], gg);
      } else {
        gg = uj * a.sinv[n * NU + n];
      }
    } else {
      gg = 0.f;
      for (int m = 0; m < NU; ++m) gg = fmaf(ac.Sm[n * NU + m], Ue[t0 + m], gg);
    }
    G[j] = a.lambda_ * gg;
  }
  __syncthreads();
  // this lane's own control dimension g: constants as scalars
  const float sd_g = a.coloured ? 1.f : a.L[g * NU + g], lo_g = a.umin[g], hi_g = a.umax[g];
  float Lrow[NU];
  if constexpr (!DIAG) {
#pragma unroll
    for (int m = 0; m < NU; ++m) Lrow[m] = ac.Lm[g * NU + m];
  }

  // ---- samples of this wave: tile i covers k = kbase + 16 i + s ----
  const int kbase = blockIdx.x * 256 + wv * (NT * 16);
  int kk[NT], orow[NT];
  bool act[NT];
  float x[NT][4], cpart[NT], ppart[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int kraw = kbase + 16 * i + s;
    act[i] = kraw < a.K;
    kk[i] = act[i] ? kraw : a.K - 1;
    orow[i] = overwrite_row(a, a.k_offset + kk[i]);
    const float* __restrict__ s0 = a.state_per_sample ? a.state + (long long)kk[i] * NX : a.state;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[i][r] = s0[4 * g + r];
    cpart[i] = 0.f;
    ppart[i] = 0.f;
  }

  // nu = 4: one row-of-4 per (timestep, sample).  Lane (g,s) needs only component g of it unless the
  // row must be coloured by a full Sigma (or is generated here): a 4-byte load of that component --
  // picking it out of a loaded float4 with a lane-dependent select makes hipcc index the row
  // dynamically, i.e. through scratch memory
  constexpr bool ROW1 = NOISE != MPPI_NOISE_PHILOX && (DIAG || NOISE == MPPI_NOISE_ACTIONS);
  float zc[NT][4], zn[NT][4];
  auto fetch = [&](int t, float (&dst)[NT][4]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if constexpr (ROW1) dst[i][0] = a.z[((long long)t * a.zp + kk[i]) * 4 + g];
      else noise4<float, NOISE == MPPI_NOISE_ACTIONS ? MPPI_NOISE_TNK4 : NOISE>(a, t, kk[i], dst[i]);
    }
  };
  fetch(0, zn);

  for (int t = 0; t < a.Tn; ++t) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int c = 0; c < (ROW1 ? 1 : 4); ++c) zc[i][c] = zn[i][c];
    }
    if constexpr (NOISE == MPPI_NOISE_PHILOX) {
      if (a.z != nullptr && g == 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
          if (act[i]) store4<float>(const_cast<float*>(a.z), a.zp, t, kk[i], zc[i]);
      }
    }
    fetch(t + 1 < a.Tn ? t + 1 : t, zn);   // prefetch the next step's rows (compute >> latency here)

    const float Ut = Ue[t * NU + g], Umt = Um[t * NU + g], Gt = G[t * NU + g];
    float ub[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float v;
      if constexpr (NOISE == MPPI_NOISE_ACTIONS) {
        v = zc[i][0];                                   // ROW1: component g was loaded
      } else if constexpr (DIAG) {
        const float zg = ROW1 ? zc[i][0] : pick4(zc[i], g);
        v = fmaf(zg, sd_g, Umt);
      } else {
        float acc = Umt;
#pragma unroll
        for (int m = 0; m < NU; ++m) acc = fmaf(zc[i][m], Lrow[m], acc);
        v = acc;
      }
      if (orow[i] == -1) v = 0.f;
      else if (orow[i] >= 0) v = a.sampler[((long long)orow[i] * a.Tn + t) * NU + g];
      v = clampT(v, lo_g, hi_g);
      const float e = v - Ut;
      ppart[i] = fmaf(Gt, ac.abs_cost ? fabsf(e) : e, ppart[i]);
      ub[i] = a.u_scale * v;
      cpart[i] = fmaf(qu_g * ub[i], ub[i], cpart[i]);
    }

    f32x4 O[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Software pipeline over the hidden tiles.  A wave issues in order, so VALU work only
    // overlaps the matrix pipe if it sits BETWEEN MFMAs in the instruction stream.  Stage 1 of
    // tile m therefore alternates one layer-1 MFMA of tile m+1 with one tanh of tile m (pinned
    // with sched_barrier so the compiler keeps that order); stage 2 is the 16 layer-2 MFMAs.
    f32x4 Hc[2][NT];
    {
      const float4 bb = *reinterpret_cast<const float4*>(b1s + 4 * g);
#pragma unroll
      for (int i = 0; i < NT; ++i) Hc[0][i] = f32x4{bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int q = 0; q < 5; ++q) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
          Hc[0][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1r[0][q], q < 4 ? x[i][q] : ub[i], Hc[0][i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < HT; ++m) {
      float th[NT][4];
      const bool more = m + 1 < HT;
      if (more) {
        const float4 bb = *reinterpret_cast<const float4*>(b1s + 16 * (m + 1) + 4 * g);
#pragma unroll
        for (int i = 0; i < NT; ++i) Hc[(m + 1) & 1][i] = f32x4{bb.x, bb.y, bb.z, bb.w};
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int slot = q * NT + i;          // 20 MFMA slots, the first 16 also carry one tanh
          if (more)
            Hc[(m + 1) & 1][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                w1r[more ? m + 1 : m][q], q < 4 ? x[i][q] : ub[i], Hc[(m + 1) & 1][i], 0, 0, 0);
          if (slot < 4 * NT) th[slot % NT][slot / NT] = half_one_minus_tanh(Hc[m & 1][slot % NT][slot / NT]);
#ifndef MPPI_MLP_NOSB
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
          O[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[m][r], th[i][r], O[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[i][r] = fmaf(rs, O[i][r] + b2r[r], x[i][r]);
        cpart[i] = fmaf(qx4[r] * x[i][r], x[i][r], cpart[i]);
      }
    }
  }

  // ---- per-sample totals: sum the 4 lane groups (dims 4g..4g+3 / control dim g) ----
  float bm = inf_v<float>();
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float c = cpart[i], p = ppart[i];
    c += __shfl_xor(c, 16, WAVE); c += __shfl_xor(c, 32, WAVE);
    p += __shfl_xor(p, 16, WAVE); p += __shfl_xor(p, 32, WAVE);
    const float total = c + p;
    if (act[i] && g == 0) {
      a.cost[kk[i]] = total;
      if (a.pert != nullptr) a.pert[kk[i]] = p;
    }
    if (act[i]) bm = fminf(bm, total);
  }
  bm = wave_min(bm);
  if (lane == 0) red[wv] = bm;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = red[0];
#pragma unroll
    for (int i = 1; i < MLP_THREADS / WAVE; ++i) r = fminf(r, red[i]);
    // workgroup covers 256 samples = 4 slots of the per-64-sample minima array
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((int)blockIdx.x * 4 + q < a.nb1) a.block_min[blockIdx.x * 4 + q] = r;
    stamp_exit(a.tstamp);
  }
}

template <int HT>
static int launch_ht(const KArgs<float>& a_in, hipStream_t st) {
  KArgs<float> a = a_in;
  const bool diag = a.diag != 0 || a.coloured != 0;
  const size_t smem = (size_t)(3 * a.J + MLP_THREADS / WAVE + HT * 16 + 2 * MLP_NU * MLP_NU) * sizeof(float);
  const dim3 grid((a.K + 255) / 256, 1, a.n_env), block(MLP_THREADS);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  profile_next_events(&ev0, &ev1, &a.tstamp);
#define MPPI_LAUNCH1(KERNEL)                                                                         \
  do {                                                                                               \
    if (ev1 != nullptr) hipExtLaunchKernelGGL(KERNEL, grid, block, smem, st, ev0, ev1, 0, a);        \
    else hipLaunchKernelGGL(KERNEL, grid, block, smem, st, a);                                       \
  } while (0)
#define MPPI_LAUNCH(NOISE_)                                                                          \
  do {                                                                                               \
    if (diag) MPPI_LAUNCH1((rollout_mlp_mfma_kernel<HT, NOISE_, true>));                             \
    else MPPI_LAUNCH1((rollout_mlp_mfma_kernel<HT, NOISE_, false>));                                 \
  } while (0)
  if (a.noise_src == MPPI_NOISE_PHILOX) MPPI_LAUNCH(MPPI_NOISE_PHILOX);
  else if (a.noise_src == MPPI_NOISE_ACTIONS) MPPI_LAUNCH(MPPI_NOISE_ACTIONS);
  else MPPI_LAUNCH(MPPI_NOISE_TNK4);
#undef MPPI_LAUNCH
#undef MPPI_LAUNCH1
  return (int)hipGetLastError();
}

bool mlp_mfma_supported(int nx, int nu, int hidden) {
  return nx == MLP_NX && nu == MLP_NU && (hidden == 64 || hidden == 128 || hidden == 256);
}

int rollout_mlp_mfma(const KArgs<float>& a, hipStream_t st) {
  if (a.mp == nullptr) return MPPI_E_BADARG;
  if (!mlp_mfma_supported(a.nx, a.nu, a.hidden) || a.states != nullptr) return MPPI_E_UNSUPPORTED;
  switch (a.hidden) {
    case 64: return launch_ht<4>(a, st);
    case 128: return launch_ht<8>(a, st);
    default: return launch_ht<16>(a, st);
  }
}

}  // namespace mppi
