// rollout_onchip.hpp -- the rng="philox" command that generates its normals where it uses them: no (K,T,nu)
// array of normals exists; since round 4 the part of the bounded noise that does not fit the chip waits in a spill array.
//
// The streaming command writes the draw once and reads it twice (generator -> K1 -> K3: 604 MB for 201.6 MB
// of algorithmic input at C3).  With the engine's own generator the normals are a pure function of
// (seed, call, sample, row), so here ONE launch does, per lane = sample,
//   G  generate the rows of the sample (Philox4x32-10 + Box-Muller, common.hpp), a batch of super-steps at a time,
//   R  roll out (mppi.py:297-332, :375-420, :186-199 -- the arithmetic of rollout_step), and KEEP the bounded
//      noise eps' = clamp(U + eps) - U (:385) ON CHIP until the sample's weight is known:
//        * the first 5 weighting tiles (300 / 320 values) in registers, 256 of them accumulation registers -- the
//          kernel runs one wave per SIMD anyway (K = 65536 is one wave per SIMD of work), the AGPR half of the
//          unified 512-entry register file is otherwise idle;
//        * the next `nsl` super-steps in LDS ([row][thread][4], one conflict-free ds_write/read_b128 per row);
//        * what does not fit waits in the caller's spill array (ABI 20, `a.spill`: [row][padded sample][4], one coalesced
//          16-byte store per lane and row behind the rollout, fetched back a weighting tile ahead in W, between the tiles that
//          come out of registers and LDS) -- or, without that array, is generated a second time in W (round 3's form;
//          13 us of generator work at C3 against ~5 us of stores + ~5 us of exposed fetches);
//   W  the workgroup's own part of K3 (mppi.py:254-259, :268): weights relative to the WORKGROUP's minimum
//      beta_b, eta_b, P_b[j] = sum_k w_k eps'_k[j] over its 256 samples -- the algebra of the single-launch
//      command (rollout.hpp FUSE block) and of the multi-GPU combine.  Column sums by a transposing wave reduction
//      on v_permlane32_swap / v_permlane16_swap / DPP (no LDS crossbar), the four waves' results combined once.
// A second, small launch (finalize_blocks_kernel, update.hip) rescales and sums the K/256 partial records in
// block order and applies K4.  Measured at C3 (tools/micro/onchip_parts.hip on THIS kernel, profiles/r03_onchip_parts.txt,
// r03_final_*; prototypes: tools/micro/k1ret_micro.hip): G alone 33.9 us (the chip-wide generator floor), + R 15, + keeping 3,
// W 29 (second generation 17.5, reduction 11.5): 81-83 us back to back, 88-90 us inside the command, against 34 + 33 + 36 us for
// generator + K1 + K3; 1.6 MB of HBM traffic per launch by PMC against 604 MB.  Round 4 (profiles/r04_onchip_spill.txt, r04_spill_*):
// 75.8 -> 70.9 us inside the command with the spill array (2 x 94 MB of traffic: 87 of the sample's 192 rows), 74.4 without it
// (the weighting phase in per-source loops).
// Scope: fp32, diagonal Sigma (a full-Sigma form -- L z + mu per timestep in the lane -- is below, behind MPPI_ONCHIP_FULL_SIGMA:
// tested, slower than generator-coloured rows), MPPI and SMPPI (base sequence, 1/dt rescaling, smoothness cost) but not KMPPI,
// M = 1, no sampler rows (the sample_null_action row is handled), no `states` output, one environment.
#pragma once
// (included from rollout.hpp, inside its include set)

namespace mppi {

template <int NU>
struct OnChip {
  static constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  // (MPPI_ONCHIP_PBROWS: common.hpp)
  static constexpr int PB = (MPPI_ONCHIP_PBROWS / P4) > 0 ? (MPPI_ONCHIP_PBROWS / P4) : 1;   // super-steps generated together (~12 rows: the Philox chains interleave)
  static constexpr int SW = (16 / P4) > 0 ? (16 / P4) : 1;       // super-steps per weighting tile
  static constexpr int TRW = SW * P4;                            // rows per tile (15 for nu = 12, else <= 16)
  static constexpr int TC = TRW * 4;                             // columns per tile (<= 64)
  // (MPPI_ONCHIP_NTA: common.hpp -- measured at C3: 4 tiles 82.8 us | 5 tiles 81.1 (214 VGPRs) | 6 tiles 80.9 with scratch, profiles/r03_onchip_parts.txt)
  static constexpr int NTA = MPPI_ONCHIP_NTA;                    // tiles kept in registers (the first 256 values: accumulation registers)
  static constexpr int AG_SS = NTA * SW, AG_ROWS = AG_SS * P4;   // 75 or 80 rows: 256 values in accumulation registers, the rest in VGPRs
  static constexpr int RG = P4 >= 3 ? 1 : (P4 == 2 ? 2 : 4);     // super-steps regenerated together (>= 3 interleaved chains)
  static constexpr bool OK = P4 <= 16 && (SW % RG) == 0;
};

// a model opts out with `static constexpr bool NO_KMPPI_FUSE = true` (the per-lane MLP: no registers to spare)
template <class Model>
constexpr bool onchip_model_ok() { return KmppiModelOk<Model>::value && OnChip<Model::NU>::OK; }

__device__ __forceinline__ float keep_in_agpr(float v) {
  float r;
  asm volatile("; eps' -> %0" : "=a"(r) : "0"(v));
  return r;
}

// The three per-(t,n) tables of K1 (rollout.hpp StepTables) padded to whole super-steps and 16-byte aligned: a
// super-step is P4 ds_read_b128 per table whatever nu is, issued together, and everything behind them is register
// arithmetic.  (Per-element reads under the `t < T` test compile to a branch, a ds_read_b32 and a full LDS round trip
// PER ELEMENT: measured 48 us of rollout instead of 15.)  Padding entries are zero.
template <int NU>
struct OnChipRow {
  float ue[Stream<NU>::P4 * 4], um[Stream<NU>::P4 * 4], g[Stream<NU>::P4 * 4];
  __device__ __forceinline__ void load(const StepTables<float>& tb, int ss, bool with_g) {
    constexpr int P4 = Stream<NU>::P4;
    const float4* __restrict__ pe = reinterpret_cast<const float4*>(__builtin_assume_aligned(tb.Ue, 16)) + ss * P4;
    const float4* __restrict__ pm = reinterpret_cast<const float4*>(__builtin_assume_aligned(tb.Um, 16)) + ss * P4;
    const float4* __restrict__ pg = reinterpret_cast<const float4*>(__builtin_assume_aligned(tb.G, 16)) + ss * P4;
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      const float4 e4 = pe[i], m4 = pm[i];
      ue[4 * i] = e4.x; ue[4 * i + 1] = e4.y; ue[4 * i + 2] = e4.z; ue[4 * i + 3] = e4.w;
      um[4 * i] = m4.x; um[4 * i + 1] = m4.y; um[4 * i + 2] = m4.z; um[4 * i + 3] = m4.w;
      if (with_g) {
        const float4 g4 = pg[i];
        g[4 * i] = g4.x; g[4 * i + 1] = g4.y; g[4 * i + 2] = g4.z; g[4 * i + 3] = g4.w;
      }
    }
  }
};

// bounded actions and noise of one super-step from its standard normals: z (in) -> eps' (out), v (out);
// exactly K1's arithmetic (rollout_step, DIAG): v = clamp(fma(z, sd, U + mu)), eps' = v - U.
// (Timesteps beyond the horizon -- the padding of the last super-step -- give values nobody reads.)
// DIAG = false: eps = L z + mu with L = chol(Sigma) out of LDS (ac.Lm), a whole timestep at a time -- "correlated
// Gaussian noise via a Cholesky-factored noise_sigma", applied in the lane that owns the sample (mppi.py:204-206).
// ESC: the bounded noise is rescaled (SMPPI: eps' = (v - B) / dt, mppi.py:544)
// `null_in_wave` (wave-uniform): some lane of the wave is the sample_null_action row
// esc (wave-uniform, in fact launch-uniform): the rescale, applied IN PLACE behind the common code -- two instantiations chosen by
// a branch per super-step left 24-66 register moves at every merge (339 v_mov_b32 in the batch loop of round 3's kernel)
template <int NU, bool DIAG>
__device__ __forceinline__ void onchip_actions(const ActionConsts<float, NU>& ac, const OnChipRow<NU>& row, int orow, bool null_in_wave,
                                               float (&z)[Stream<NU>::P4 * 4], float (&v)[Stream<NU>::P4 * 4], bool esc) {
  if constexpr (DIAG) {
#pragma unroll
    for (int f = 0; f < Stream<NU>::P4 * 4; ++f) v[f] = fmaf(z[f], ac.sd[f % NU], row.um[f]);   // mppi.py:201-206, :380
  } else {
#pragma unroll
    for (int tt = 0; tt < Stream<NU>::TT; ++tt) {
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        float Lr[NU];
        chol_row<float, NU>(ac.Lm, n, Lr);
        float s = row.um[tt * NU + n];
#pragma unroll
        for (int m = 0; m <= n; ++m) s = fmaf(z[tt * NU + m], Lr[m], s);                         // L is lower triangular
        v[tt * NU + n] = s;
      }
    }
  }
  if (null_in_wave) {                                                   // mppi.py:390-392: the only wave that pays for the select
#pragma unroll
    for (int f = 0; f < Stream<NU>::P4 * 4; ++f) v[f] = orow == -1 ? 0.f : v[f];
  }
#pragma unroll
  for (int f = 0; f < Stream<NU>::P4 * 4; ++f) {
    const int n = f % NU;
    float w = v[f];
    w = clampT(w, ac.lo[n], ac.hi[n]);                                  // :383
    v[f] = w;
    z[f] = w - row.ue[f];                                               // :385
  }
  if (esc) {
#pragma unroll
    for (int f = 0; f < Stream<NU>::P4 * 4; ++f) z[f] *= ac.e_scale;    // SMPPI :544
  }
}

// plain (wave-uniform): no |noise| cost, u_scale == 1, no SMPPI terms -- the common case; the extras are applied in place under
// the flag (one code path: no merge copies)
template <class Model>
__device__ __forceinline__ void onchip_steps(const KArgs<float>& a, const ActionConsts<float, Model::NU>& ac, const Model& model,
                                             const OnChipRow<Model::NU>& row, int ss, const float (&e)[Stream<Model::NU>::P4 * 4],
                                             const float (&v)[Stream<Model::NU>::P4 * 4], float (&x)[Model::NX], float (&vprev)[Model::NU],
                                             float& rollout, float& pert, bool plain) {
  constexpr int NU = Model::NU, TT = Stream<NU>::TT;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int t = ss * TT + tt;
    if (t < a.Tn) {
      float u[NU], en[NU];
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        u[n] = v[tt * NU + n];
        en[n] = e[tt * NU + n];
      }
      if (!plain) {
        if (a.smooth_w != 0.f) {
          // SMPPI smoothness cost w * |u_scale * (v[t] - v[t-1])|^2 (mppi.py:559-562), as rollout_step has it
          float d2 = 0.f;
#pragma unroll
          for (int n = 0; n < NU; ++n) {
            const float d = u[n] - vprev[n];
            d2 = fmaf(d, d, d2);
            vprev[n] = u[n];
          }
          if (t > 0) rollout = fmaf(a.smooth_w, d2, rollout);
        }
#pragma unroll
        for (int n = 0; n < NU; ++n) {
          en[n] = ac.abs_cost ? fabsf(en[n]) : en[n];
          u[n] *= a.u_scale;                                             // :313
          // (the scaled action is a value of its own: left to -ffp-contract=fast, `x + u * u_scale` in the model's step became one
          //  fma in one on-chip kernel and a multiply and an add in the other -- an ulp of the action apart, round 6)
          asm volatile("" : "+v"(u[n]));
        }
      }
#pragma unroll
      for (int n = 0; n < NU; ++n) pert = fmaf(row.g[tt * NU + n], en[n], pert);   // :409, :415
      model.step(x, u, t);                                              // :314
      rollout += model.cost(x, u, t);                                   // :318-319
    }
  }
}

// SEVEN: Philox4x32-7 instead of -10 (rng="philox7") -- a template parameter here: as a run-time (wave-uniform) branch inside the twelve
// interleaved generator chains it cost the kernel 10 spilled VGPRs (44 B of scratch)
#if defined(MPPI_ONCHIP_EXP) && (MPPI_ONCHIP_EXP & 16)   // experiment: two workgroups per CU (with bits 1 | 4: what a second wave per SIMD buys G + R)
#define MPPI_ONCHIP_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define MPPI_ONCHIP_OCC
#endif
template <class Model, bool DIAG, bool SEVEN = false>
__global__ void __launch_bounds__(K1_BLOCK) MPPI_ONCHIP_OCC rollout_onchip_kernel(const KArgs<float> a, const int nsl, const int nsm) {
  using T = float;
  constexpr int NX = Model::NX, NU = Model::NU;
  using OC = OnChip<NU>;
  constexpr int P4 = OC::P4, TT = OC::TT, PB = OC::PB, SW = OC::SW, TRW = OC::TRW, TC = OC::TC, NTA = OC::NTA, AG_SS = OC::AG_SS,
                RG = OC::RG;
  static_assert(K1_BLOCK == 256 && K1_BLOCK == BLOCK, "four waves per workgroup; onchip_carve (common.hpp) counts BLOCK-sample records");
  stamp_entry(a.tstamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nss = (a.Tn + TT - 1) / TT;
  const int ntiles = (nss + SW - 1) / SW;
  const int Jp = nss * P4 * 4;
  const OnChipLds L{Jp, ntiles, nsl, P4, DIAG ? 0 : 2 * NU * NU};
  T* Ue = reinterpret_cast<T*>(smem_raw);
  T* Um = Ue + Jp;
  T* G = Um + Jp;
  T* red = G + Jp;
  T* fac = red + 4;                      // [2*NU*NU] chol(Sigma) | Sigma^-1 (full Sigma only), 16-byte aligned
  T* ex = Ue + L.tables();
  float4* keepL = reinterpret_cast<float4*>(ex + L.ex());
  for (int j = threadIdx.x; j < Jp; j += K1_BLOCK) {
    const bool in = j < a.J;
    const int n = j % NU;
    const T u = in ? u_base(a, j) : T(0);         // what the noise is added to and measured from (SMPPI: A + U dt)
    Ue[j] = u;
    Um[j] = in ? u + a.mu[n] : T(0);
    if constexpr (DIAG) G[j] = in ? a.lambda_ * ((a.B != nullptr ? u_eff(a, j) : u) * a.sinv[n * NU + n]) : T(0);   // always the true U
  }
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int kraw = blockIdx.x * K1_BLOCK + threadIdx.x;
  const bool active = kraw < a.K;
  const int k = active ? kraw : a.K - 1;
  const int orow = overwrite_row(a, a.k_offset + k);
  const Model model(a);
  T x[NX];
  {
    const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = s0[i];                          // mppi.py:302-305
  }
  ActionConsts<T, NU> ac;
  ac.load(a, DIAG ? nullptr : fac);
  float keepA[OC::AG_ROWS * 4];
#pragma unroll
  for (int i = 0; i < OC::AG_ROWS * 4; ++i) keepA[i] = i < 256 ? keep_in_agpr(0.f) : 0.f;   // rows beyond a short horizon read as zero noise
  __syncthreads();
  if constexpr (!DIAG) {
    // full Sigma: G[t,n] = lambda * sum_m Sigma^-1[n,m] U[t,m] needs the whole row of U and the factor block -> second pass
    for (int j = threadIdx.x; j < Jp; j += K1_BLOCK) {
      const int n = j % NU, t0 = j - n;
      T g = T(0);
      if (j < a.J) {
        for (int m = 0; m < NU; ++m) g = fmaf(ac.Sm[n * NU + m], a.B != nullptr ? u_eff(a, t0 + m) : Ue[t0 + m], g);   // Sigma^-1 symmetric
      }
      G[j] = a.lambda_ * g;
    }
    __syncthreads();
  }
  const StepTables<T> tb{Ue, Um, G, nullptr, kraw - lane};
  const long long kg = a.k_offset + k;
  // super-steps [M0, M1) wait in memory (a.spill: [row][padded sample][4], one coalesced 16-byte store / load per lane and row)
  const int M0 = AG_SS + nsl, M1 = M0 + nsm;
  float4* __restrict__ spill4 = reinterpret_cast<float4*>(a.spill) + kraw;
  const long long Kp = (long long)gridDim.x * K1_BLOCK;
  const bool null_in_wave = __any(orow == -1);
  const bool plain = !a.abs_cost && a.u_scale == 1.f && a.e_scale == 1.f && a.smooth_w == 0.f;
  T vprev[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) vprev[n] = T(0);

  // ---- G + R: a batch of PB super-steps at a time ----
  T rollout = 0.f, pert = 0.f;
  constexpr int NCA = (AG_SS + PB - 1) / PB;     // batches that touch the accumulation registers (static indices)
  for (int bi = 0; bi * PB < nss; ++bi) {
    T zb[PB][P4 * 4], vb[PB][P4 * 4];
#pragma unroll
    for (int b = 0; b < PB; ++b) {
#pragma unroll
      for (int i = 0; i < P4; ++i) {
        T r[4];
        philox_normal4<T, SEVEN ? 7 : 10>(a.seed, a.call, kg, (long long)(bi * PB + b) * P4 + i, r);   // rows past the horizon: unused
        zb[b][4 * i + 0] = r[0]; zb[b][4 * i + 1] = r[1]; zb[b][4 * i + 2] = r[2]; zb[b][4 * i + 3] = r[3];
      }
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const int ss = bi * PB + b;
#if defined(MPPI_ONCHIP_EXP) && (MPPI_ONCHIP_EXP & 8)    // experiment (tools/micro/onchip_parts.hip): no rollout arithmetic
#pragma unroll
      for (int q = 0; q < P4 * 4; ++q) rollout += zb[b][q];
      continue;
#endif
      OnChipRow<NU> row;
      row.load(tb, ss < nss ? ss : nss - 1, true);
      onchip_actions<NU, DIAG>(ac, row, orow, null_in_wave, zb[b], vb[b], !plain);
      onchip_steps<Model>(a, ac, model, row, ss, zb[b], vb[b], x, vprev, rollout, pert, plain);
    }
#if defined(MPPI_ONCHIP_EXP) && (MPPI_ONCHIP_EXP & 4)      // experiment: nothing kept
    continue;
#endif
    // keep eps': accumulation registers (static index -> a chain of uniform compares over the batch), then LDS
    auto keep_lds = [&](int b) {
      const int ss = bi * PB + b;
      if (ss >= AG_SS && ss < M0) {
        asm volatile("; keep in LDS");          // (distinct markers: merged into one store through a generic pointer the two branches
#pragma unroll                                  //  become flat_store instructions, which wait on both memory counters)
        for (int i = 0; i < P4; ++i)
          keepL[((ss - AG_SS) * P4 + i) * K1_BLOCK + threadIdx.x] = make_float4(zb[b][4 * i], zb[b][4 * i + 1], zb[b][4 * i + 2], zb[b][4 * i + 3]);
        asm volatile("; kept in LDS" ::: "memory");
      } else if (ss >= M0 && ss < M1) {
        asm volatile("; keep in memory");
#pragma unroll
        for (int i = 0; i < P4; ++i)
          spill4[(long long)((ss - M0) * P4 + i) * Kp] = make_float4(zb[b][4 * i], zb[b][4 * i + 1], zb[b][4 * i + 2], zb[b][4 * i + 3]);
        asm volatile("; kept in memory" ::: "memory");
      }
    };
    bool done = false;
    static_for<0, NCA>([&](auto cc) {
      constexpr int C = decltype(cc)::value;
      if (bi == C) {
        static_for<0, PB>([&](auto bb) {
          constexpr int B = decltype(bb)::value, SS = C * PB + B;
          if constexpr (SS < AG_SS) {
#pragma unroll
            for (int q = 0; q < P4 * 4; ++q) keepA[SS * P4 * 4 + q] = (SS * P4 * 4 + q) < 256 ? keep_in_agpr(zb[B][q]) : zb[B][q];
          } else {
            keep_lds(B);
          }
        });
        done = true;
      }
    });
    if (!done) {
#pragma unroll
      for (int b = 0; b < PB; ++b) keep_lds(b);
    }
  }
  if (a.use_terminal) rollout += model.terminal(x);                     // :324-328
  const T total = rollout + pert;                                       // :416
  if (active) {
    a.cost[k] = total;
    if (a.pert != nullptr) a.pert[k] = pert;
  }

#if defined(MPPI_ONCHIP_EXP) && (MPPI_ONCHIP_EXP & 1)      // experiment: generate + roll out only
  return;
#endif
  // ---- W: this workgroup's partial record {beta_b, eta_b, P_b} ----
  const T inv_lambda = T(1) / a.lambda_;
  const T beta_b = block_min<T>(active ? total : inf_v<T>(), red);
  const T wk = active ? weight_of<T>(total, beta_b, inv_lambda) : T(0);
  const T eta_b = block_sum<T>(wk, red);
  // a wave whose weights are all EXACTLY zero (fp32 exp underflow: a peaked softmax) adds exactly nothing
  const bool live = __ballot(wk != T(0)) != 0ull;
  // Tiles by where their rows waited, each kind in its own loop (what is live in one -- the kept registers, the generator's
  // constants -- is dead in the others; one loop over all tiles with the source chosen inside spilled 26 VGPRs):
  //   A  [0, tA)      kept in registers; the tile that holds the VGPR-kept values goes first (its registers take the fetched rows)
  //   B  [tA, tB)     kept in LDS                          (nsl is a whole number of tiles)
  //   C  [tB, tC)     waited in memory                     (nsm is a whole number of tiles; rows past the horizon read as zero)
  //   D  [tC, ntiles) generated a second time              (the rest: memory and the generator work side by side)
  // The memory tiles are not a loop of their own: ONE of them is in flight at any time, fetched into `pf`, and it is consumed --
  // and the next one fetched -- between the tiles of A, B and D once ~3 us of their arithmetic has gone by (`work`: a tile of
  // column sums counts 1, a tile generated again 3; a fetch of 15.7 MB chip-wide takes ~3.3 us).  Measured at C3: everything
  // beyond LDS generated again 76 us, everything fetched behind a loop of its own 77 us (the chip idles on HBM for 20 us), interleaved
  // see profiles/r04_onchip_spill.txt.
  const int tA = ntiles < NTA ? ntiles : NTA;
  const int tB = tA + nsl / SW < ntiles ? tA + nsl / SW : ntiles;
  const int tC = tB + nsm / SW < ntiles ? tB + nsm / SW : ntiles;
  auto column_sums = [&](int tile, const T (&e)[TRW][4]) __attribute__((always_inline)) {
    T acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = i < TC ? wk * e[(i < TC ? i : 0) / 4][i % 4] : T(0);
    ex[(wv * ntiles + tile) * 64 + lane] = wave_reduce_transpose64<float>(acc);
  };
  float4 pf[TRW];
  int mt = tB, work = 0;                                                 // the memory tile in flight (mt < tC), arithmetic since its fetch
  auto fetch = [&](int tile) __attribute__((always_inline)) {                                           // tile in [tB, tC): super-steps M0 + (tile - tB) SW ...
#pragma unroll
    for (int i = 0; i < TRW; ++i) pf[i] = spill4[(long long)((tile - tB) * TRW + i) * Kp];
  };
  auto consume = [&]() __attribute__((always_inline)) {                                                 // the tile in flight -> column sums; the next one takes off
    T acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int r = (i < TC ? i : 0) / 4, q = i % 4;
      const float v = q == 0 ? pf[r].x : q == 1 ? pf[r].y : q == 2 ? pf[r].z : pf[r].w;
      acc[i] = i < TC ? (mt * SW + r / P4 < nss ? wk * v : T(0)) : T(0);
    }
    const int tile = mt++;
    if (mt < tC) fetch(mt);
    ex[(wv * ntiles + tile) * 64 + lane] = wave_reduce_transpose64<float>(acc);
    work = 0;
  };
  auto from_registers = [&](int tile) __attribute__((always_inline)) {
    T e[TRW][4];
    static_for<0, NTA>([&](auto tt) {
      constexpr int TI = decltype(tt)::value;
      if (tile == TI) {
        asm volatile("; register tile %0" ::"n"(TI));                    // (distinct per branch: the branches must not be merged into
#pragma unroll                                                           //  one copy indexed by `tile` -- that puts keepA in scratch)
        for (int i = 0; i < TRW; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) e[i][q] = keepA[(TI * TRW + i) * 4 + q];
      }
    });
    column_sums(tile, e);
  };
  if (!live) {
    for (int tile = 0; tile < ntiles; ++tile) ex[(wv * ntiles + tile) * 64 + lane] = T(0);
  } else {
    if (tA == NTA) from_registers(NTA - 1);                              // A: the tile with the VGPR-kept values
    if (mt < tC) fetch(mt);
    for (int tile = 0; tile < (tA == NTA ? NTA - 1 : tA); ++tile) {      // A: the rest
      from_registers(tile);
      if (++work >= 3 && mt < tC) consume();
    }
    for (int tile = tA; tile < tB; ++tile) {                             // B
      T e[TRW][4];
#pragma unroll
      for (int i = 0; i < TRW; ++i) {
        const float4 q4 = keepL[((tile - tA) * TRW + i) * K1_BLOCK + threadIdx.x];
        e[i][0] = q4.x; e[i][1] = q4.y; e[i][2] = q4.z; e[i][3] = q4.w;
      }
      column_sums(tile, e);
      if (++work >= 3 && mt < tC) consume();
    }
    // (the generator's sample index is made up again here instead of living in two registers through the whole kernel: the
    //  LinearGoal<12,4> instantiation spilled exactly those two)
    int kd = blockIdx.x * K1_BLOCK + threadIdx.x;
    asm volatile("" : "+v"(kd));
    const long long kgd = a.k_offset + (kd < a.K ? kd : a.K - 1);
    for (int tile = tC; tile < ntiles; ++tile) {                         // D
      T e[TRW][4];
#pragma unroll
      for (int g = 0; g < SW / RG; ++g) {
        const int ss0 = tile * SW + g * RG;
#if defined(MPPI_ONCHIP_EXP) && (MPPI_ONCHIP_EXP & 2)      // experiment: no second generation (zeros instead)
        if (ss0 < 0) {
#else
        if (ss0 < nss) {
#endif
          T zg[RG][P4 * 4], vg[P4 * 4];
#pragma unroll
          for (int s = 0; s < RG; ++s)
#pragma unroll
            for (int i = 0; i < P4; ++i) {
              T r[4];
              philox_normal4<T, SEVEN ? 7 : 10>(a.seed, a.call, kgd, (long long)(ss0 + s) * P4 + i, r);
              zg[s][4 * i + 0] = r[0]; zg[s][4 * i + 1] = r[1]; zg[s][4 * i + 2] = r[2]; zg[s][4 * i + 3] = r[3];
            }
#pragma unroll
          for (int s = 0; s < RG; ++s) {
            OnChipRow<NU> row;
            row.load(tb, (ss0 + s) < nss ? ss0 + s : nss - 1, false);
            onchip_actions<NU, DIAG>(ac, row, orow, null_in_wave, zg[s], vg, !plain);
#pragma unroll
            for (int i = 0; i < P4; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) e[(g * RG + s) * P4 + i][q] = (ss0 + s) < nss ? zg[s][4 * i + q] : T(0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < RG * P4; ++i) e[g * RG * P4 + i][0] = e[g * RG * P4 + i][1] = e[g * RG * P4 + i][2] = e[g * RG * P4 + i][3] = T(0);
        }
      }
      column_sums(tile, e);
      work += 3;
      if (mt < tC) consume();
    }
    while (mt < tC) consume();                                           // C: what the other tiles' arithmetic did not cover
  }
  __syncthreads();
  // one combine over the four waves, in wave order; column j = tile * TC + c
  for (int idx = threadIdx.x; idx < ntiles * 64; idx += K1_BLOCK) {
    const int tile = idx >> 6, c = idx & 63, j = tile * TC + c;
    if (c < TC && j < a.Jpad) {
      const int o = tile * 64 + c;
      a.P_part[(long long)blockIdx.x * a.Jpad + j] = (ex[o] + ex[ntiles * 64 + o]) + (ex[2 * ntiles * 64 + o] + ex[3 * ntiles * 64 + o]);
    }
  }
  if (threadIdx.x == 0) {
    a.eta_part[blockIdx.x] = eta_b;
    a.block_min[blockIdx.x] = beta_b;
  }
  if (a.tstamp != nullptr) {
    __syncthreads();
    stamp_exit(a.tstamp);
  }
}

template <typename T>
static bool onchip_problem_ok(const KArgs<T>& a) {
  static const int off = [] { const char* e = getenv("MPPI_ONCHIP"); return e ? atoi(e) == 0 : 0; }();
  return !off && sizeof(T) == 4 && a.fuse >= 0 && a.noise_src == MPPI_NOISE_PHILOX && a.z == nullptr && !a.coloured &&
         a.M == 1 && a.n_env == 1 && a.n_sampler == 0 && a.states == nullptr && a.W == nullptr && a.record != nullptr &&
         (a.fuse == 0 || a.U_out != nullptr) &&
         (a.K + K1_BLOCK - 1) / K1_BLOCK <= 8192;
}

// Models whose seven-round instantiation (rng="philox7") the register allocator does not fit into the 512 registers of the kernel
// (tests/test_kernel_resources.py reads the code objects): those keep the rows-in-memory forms under philox7.  LinearGoal (12, 4):
// 44 spilled VGPRs with SEVEN = true, none with ten rounds (the shorter chains are scheduled differently).
template <class Model>
struct onchip_seven_ok : std::true_type {};
template <>
struct onchip_seven_ok<LinearGoalModel<float, 12, 4>> : std::false_type {};

template <class Model>
static int launch_rollout_onchip_pair(const KArgs<float>& a, hipStream_t st);     // rollout_onchip_pair.hpp
static inline int onchip_pair_mode();

// returns MPPI_OK_ONCHIP when the launch was issued, a positive HIP error, or -1: not this path
template <class Model, typename T>
static int launch_rollout_onchip(const KArgs<T>& a_in, hipStream_t st) {
  if constexpr (!std::is_same<T, float>::value || !onchip_model_ok<Model>()) {
    return -1;
  } else {
    if (!onchip_problem_ok(a_in)) return -1;
    if (a_in.seven && !onchip_seven_ok<Model>::value) return -1;
#ifndef MPPI_ONCHIP_FULL_SIGMA
    // The full-Sigma form (DIAG = false: L z + mu per timestep in the lane) is written and was tested at full size, but it is
    // LDS-issue-bound on the factor rows -- 0.127 ms at C3 against 0.104 ms for rows coloured by the generator launch and
    // streamed (profiles/r03_variants_philox.txt) -- and doubled this header's compile time in every model unit: not
    // instantiated unless the build defines MPPI_ONCHIP_FULL_SIGMA.
    if (a_in.diag == 0) return -1;
#endif
    constexpr int NU = Model::NU;
    using OC = OnChip<NU>;
    KArgs<T> a = a_in;
    onchip_carve(a);
    if (onchip_pair_mode() != 0) {
      const int rp = launch_rollout_onchip_pair<Model>(a, st);          // two waves per sample group where that form applies
      if (rp != -1) return rp;
    }
    const bool diag = a.diag != 0;
    const OnChipGeometry g = onchip_geometry(NU, a.Tn, diag);
    static_assert(OC::OK, "onchip_model_ok");
    if (!g.ok || g.P4 != OC::P4 || g.TT != OC::TT || g.SW != OC::SW || g.AG_SS != OC::AG_SS) return -1;
    const int nsl = g.nsl;
    // what fits neither registers nor LDS waits in a.spill when the caller provides one (else it is generated a second time)
    static const int spill_max = [] { const char* e = getenv("MPPI_ONCHIP_SPILL_SS"); return e ? atoi(e) : 1 << 30; }();
    int nsm = 0;
    if (a.spill != nullptr && spill_max > 0) {
      const long long rows = a.spill_cap / ((long long)a.nkc * K1_BLOCK * 4);
      nsm = g.nsm;
      if (nsm > rows / OC::P4) nsm = (int)(rows / OC::P4);
      if (nsm > spill_max) nsm = spill_max;
      nsm -= nsm % OC::SW;
    }
#if defined(MPPI_ONCHIP_EXP) && (MPPI_ONCHIP_EXP & 16)
    const size_t smem = OnChipLds{g.nss * g.P4 * 4, g.ntiles, 0, g.P4, 0}.bytes();   // tables + exchange only: two workgroups fit a CU
#else
    const size_t smem = g.smem;
#endif
    const dim3 grid(a.nkc), block(K1_BLOCK);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    profile_next_events(&ev0, &ev1, &a.tstamp);
#define MPPI_ONCHIP_LAUNCH(KERNEL)                                                                                  \
  do {                                                                                                              \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (ev1 != nullptr) hipExtLaunchKernelGGL(KERNEL, grid, block, smem, st, ev0, ev1, 0, a, nsl, nsm);             \
    else hipLaunchKernelGGL(KERNEL, grid, block, smem, st, a, nsl, nsm);                                            \
  } while (0)
    if (diag && a.seven) {
      if constexpr (onchip_seven_ok<Model>::value) MPPI_ONCHIP_LAUNCH((rollout_onchip_kernel<Model, true, true>));
    } else if (diag) MPPI_ONCHIP_LAUNCH((rollout_onchip_kernel<Model, true, false>));
#ifdef MPPI_ONCHIP_FULL_SIGMA
    else MPPI_ONCHIP_LAUNCH((rollout_onchip_kernel<Model, false, false>));
#endif
#undef MPPI_ONCHIP_LAUNCH
    const int e = (int)hipGetLastError();
    return e != 0 ? e : MPPI_OK_ONCHIP;
  }
}

}  // namespace mppi
