// rollout_kmppi.hpp -- K1 for KMPPI with the kernel interpolation INSIDE the launch (mppi.py:653-670):
//   theta'[s] = clamp(theta[s] + eps_S[s])               bounded control points           :657-663
//   v[t]      = sum_s W[t,s] theta'[s]                    W = K(Hs,Tk) Ktktk^-1, (T,S)     :665, :621-628
//   v[t]      = sampler row | 0, clamp, e = v - U[t], action cost, dynamics, running cost  :666-670, :297-332
// The two-launch form (mppi_kmppi_interp, then K1 on raw actions) writes and re-reads a (K,T,nu)
// array: 62 + 35 us at K = 65536, T = 64, nu = 12, S = 32 (fused: 73-76 us).  Here a lane keeps the S*nu
// bounded control points of ITS sample resident for the whole horizon (up to 256 in the accumulation-register
// half of the file, which the matrix instruction reads directly, the rest in LDS) and produces four timesteps
// at a time with v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 outer products, one per group of four lanes:
//     D[i] of lane l  +=  A(lane 4*(l/4) + i) * B(lane l)
// A = W[t0 + (l & 3)][s]   (the lane's row of the operator tile, read from LDS)
// B = theta'[s][n] of the lane's own sample
// so D[i] is v[t0 + i][n] of the lane's own sample: no exchange between lanes, nothing staged, and the rollout
// consumes the four timesteps straight out of the accumulators.  The instruction is an exact fp32 multiply-add
// (K = 1), the sum over s runs in index order.
// Cost at C3-sized work (tools/micro/kmppi_k1_parts.hip): the K*T*S*nu fp32 MACs take 22 us on the matrix pipe
// (8 cycles per instruction, 32 MAC / cycle / SIMD), the support-point rows 17 us to read (or ~20 us to generate in
// the prologue), the rollout arithmetic 13-18 us; the phases of a wave run one after the other (one wave per SIMD).
// fp32, diagonal Sigma, nu in {4, 8, 12, 16}, S*nu <= 384; anything else keeps the two-launch form.
#pragma once

namespace mppi {

typedef float kf32x4_t __attribute__((ext_vector_type(4)));

template <int NU>
struct KmppiFuse {
  static constexpr bool OK = NU % 4 == 0 && NU >= 4 && NU <= 16;
  static constexpr int SRAW = OK ? (384 / NU) & ~3 : 4;
  static constexpr int SMAX = SRAW > 64 ? 64 : SRAW;     // nu = 4: 64 | 8: 48 | 12: 32 | 16: 24
};

// a model opts out with `static constexpr bool NO_KMPPI_FUSE = true`
template <class Model, typename = void>
struct KmppiModelOk : std::true_type {};
template <class Model>
struct KmppiModelOk<Model, std::enable_if_t<Model::NO_KMPPI_FUSE>> : std::false_type {};

// Where the S*nu bounded control points of a lane live (flat index i = s*nu + n, compile-time):
//   i <  NA : accumulation registers (AGPRs) -- the matrix instruction reads its B operand straight from
//             there, so they cost the VALU nothing after the one write that puts them there;
//   i >= NA : LDS, [i/4][thread][4] (one conflict-free ds_read_b128 per four values and tile).
// The arch-VGPR half of the file stays free for the rollout itself (state, actions, the 4 x nu tile).
template <int NU>
struct KmppiRegs {
  static constexpr int NA = 256;                                   // values kept in AGPRs
  static constexpr int TOT = KmppiFuse<NU>::SMAX * NU;
  static constexpr int NL = TOT > NA ? TOT - NA : 0;               // values kept in LDS (per lane)
};

// put a value into the accumulation-register file (the compiler emits the v_accvgpr_write and knows its hazards)
__device__ __forceinline__ float to_agpr(float v) {
  float r;
  asm volatile("; theta' -> %0" : "=a"(r) : "0"(v));
  return r;
}

// ... and read it back for the VALU through an explicit copy: the value the matrix instructions read stays in the accumulation
// class for its whole life (left to the allocator, ONE live range that ends in a VALU use is given a VGPR from the start -- and
// 256 control points no longer fit beside the rollout: 250 spilled registers around the horizon loop)
__device__ __forceinline__ float from_agpr(float a) {
  float r;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(a));
  return r;
}

// the four timesteps of one accumulator tile (D[n][i] = raw action n of timestep t0 + i)
template <class Model, int MODE>
__device__ __forceinline__ void kmppi_steps4(const KArgs<float>& a, const ActionConsts<float, Model::NU>& ac, const Model& model,
                                             const StepTables<float>& tb, int k, bool active, int orow, int t0,
                                             const kf32x4_t (&D)[Model::NU], float (&x)[Model::NX], float (&vprev)[Model::NU],
                                             float& rollout, float& pert) {
  constexpr int NU = Model::NU;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + i;
    if (t < a.Tn) {
      float zt[NU];
#pragma unroll
      for (int n = 0; n < NU; ++n) zt[n] = D[n][i];
#if defined(MPPI_KMPPI_EXP) && (MPPI_KMPPI_EXP & 4)   // experiment (tools/micro/kmppi_k1_parts.hip): no rollout arithmetic
#pragma unroll
      for (int n = 0; n < NU; ++n) rollout += zt[n];
      continue;
#endif
      rollout_step<Model, float, MPPI_NOISE_ACTIONS, true, MODE, false>(a, ac, model, tb, k, active, orow, t, zt, x, vprev,
                                                                        rollout, pert);   // KMPPI: no SMPPI terms
    }
  }
}

template <class Model, int NOISE>
__global__ void __launch_bounds__(K1_BLOCK) rollout_kmppi_kernel(const KArgs<float> a) {
  constexpr int NX = Model::NX, NU = Model::NU, P4 = NU / 4, SMAX = KmppiFuse<NU>::SMAX;
  constexpr int NA = KmppiRegs<NU>::NA;
  stamp_entry(a.tstamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int S = a.S, Thor = a.Tn;
  const int T4 = (Thor + 3) & ~3, S4 = (S + 3) & ~3;        // support points beyond S only ever meet zero columns of W
  const int SPW = S4 + 4;                   // padded operator row: the four rows of a tile fall on different banks
  float* Ue = reinterpret_cast<float*>(smem_raw);       // [J] nominal sequence (what the noise is measured from)
  float* G = Ue + ((a.J + 3) & ~3);                     // [J] lambda * U / sigma^2
  float* thl = G + ((a.J + 3) & ~3);                    // [S*NU] control points theta + noise mean
  float* Wl = thl + ((S * NU + 3) & ~3);                // [T4][SPW], zero outside (T, S)
  // in-kernel theta update (a.kw): theta itself (the noise is measured from it: mppi.py:664), block reduction scratch and the
  // four waves' column sums; the bounded control points beyond the AGPRs come last
  constexpr int NWT = (SMAX * NU + 63) / 64;            // 64-column tiles of the control-point sequence
  float* th0 = Wl + T4 * SPW;                           // [S*NU]
  float* red = th0 + (a.kw ? NWT * 64 : 0);             // [4]   (th0 is read in whole tiles)
  float* exw = red + (a.kw ? 4 : 0);                    // [4 waves][NWT][64]
  float* thx = exw + (a.kw ? 4 * NWT * 64 : 0);         // [NL/4][K1_BLOCK][4] bounded control points beyond the AGPRs
  for (int j = threadIdx.x; j < a.J; j += K1_BLOCK) {
    const int n = j % NU;
    Ue[j] = u_base(a, j);
    G[j] = a.lambda_ * (u_eff(a, j) * a.sinv[n * NU + n]);
  }
  for (int i = threadIdx.x; i < S * NU; i += K1_BLOCK) {
    thl[i] = a.theta[i] + (a.coloured ? 0.f : a.mu[i % NU]);   // theta + mu
    if (a.kw) th0[i] = a.theta[i];
  }
  for (int i = threadIdx.x; i < T4 * SPW; i += K1_BLOCK) {
    const int t = i / SPW, s = i - t * SPW;
    Wl[i] = (t < Thor && s < S) ? a.W[(long long)t * S + s] : 0.f;
  }
  const Model model(a);
  ActionConsts<float, NU> ac;
  ac.load(a, nullptr);
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  kf32x4_t* thx_me = reinterpret_cast<kf32x4_t*>(thx) + threadIdx.x;     // + (i - NA)/4 * K1_BLOCK
  const int nchunks = (a.K + K1_BLOCK - 1) / K1_BLOCK;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int kraw = chunk * K1_BLOCK + threadIdx.x;
    const bool active = kraw < a.K;
    const int k = active ? kraw : a.K - 1;              // tail lanes shadow the last sample, never store
    const int orow = overwrite_row(a, a.k_offset + k);
    StepTables<float> tb{Ue, Ue, G, nullptr, kraw - lane};
    float x[NX];
    {
      const float* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = s0[i];        // mppi.py:302-305
    }
    // ---- bounded control points of this sample (mppi.py:657-663) ----
    float tha[NA < SMAX * NU ? NA : SMAX * NU];
    // (template recursion, not `#pragma unroll`: the Philox body is past the pragma's size budget, and a
    // loop left rolled would index tha[] dynamically -- i.e. put it in scratch)
    constexpr int ROWS = SMAX * P4;                     // rows-of-4 of the support-point stream per sample
    constexpr int RD = ROWS < 32 ? ROWS : 32;           // rows in flight per lane (rows from memory)
    const int last_row = S * P4 - 1;
    kf32x4_t ring[RD];
    // Rows from memory: a register ring, refilled in place, NO branch between two loads -- vector memory
    // retires in order behind one counter, and a branch between a load and its use turns every wait into
    // vmcnt(0): 96 serialised round trips (measured: 60 us of a 91 us kernel).  Rows beyond S are clamped
    // to the last row and zeroed by a select.
    auto fetch = [&](int row) {
      float r[4];
#if defined(MPPI_KMPPI_EXP) && (MPPI_KMPPI_EXP & 1)   // experiment (tools/micro/kmppi_k1_parts.hip): no row traffic
      r[0] = r[1] = r[2] = r[3] = 0.25f * (float)((row + k) & 7);
#else
      noise4<float, NOISE, 10>(a, (long long)(row < last_row ? row : last_row), k, r);   // (ten rounds at compile time: launch_rollout_kmppi refuses seven-round in-kernel generation)
#endif
      return kf32x4_t{r[0], r[1], r[2], r[3]};
    };
    if constexpr (NOISE != MPPI_NOISE_PHILOX) {
#pragma unroll
      for (int d = 0; d < RD; ++d) ring[d] = fetch(d);
    }
    static_for<0, ROWS>([&](auto ic) {
      constexpr int row = decltype(ic)::value, s = row / P4, q = row % P4;
      constexpr int i0 = s * NU + 4 * q;
      kf32x4_t z4;
      if constexpr (NOISE == MPPI_NOISE_PHILOX) {
        // generated rows: one at a time (96 independent Philox chains interleaved by the scheduler would
        // need more registers than the lane has left)
        z4 = fetch(row);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        z4 = ring[row % RD];
        if constexpr (row + RD < ROWS) ring[row % RD] = fetch(row + RD);
      }
      // theta' = clamp(theta + mu + sd z): one fma, max, min per value.  A support point beyond S gets some
      // finite value (row and theta index are clamped) that only ever meets the zero columns of the operator
      kf32x4_t v4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = 4 * q + i;
        v4[i] = clampT(m_fma(z4[i], ac.sd[n], thl[(s < S ? s : 0) * NU + n]), ac.lo[n], ac.hi[n]);
      }
      if constexpr (i0 < NA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tha[i0 + i] = to_agpr(v4[i]);
      } else {
        if (s < S4) thx_me[((i0 - NA) >> 2) * K1_BLOCK] = v4;    // own slot: no barrier, nobody else reads it
      }
    });
    // ---- four timesteps at a time: interpolate, then roll them out ----
    float rollout = 0.f, pert = 0.f;
    float vprev[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) vprev[n] = 0.f;
    // wave-uniform choice, as in K1: 0 = no overwritten row in this wave (no select, no conditional memory traffic
    // in the steps); 2 = null-action / sampler rows or a `states` output.  (K1's select-only mode 1 is not
    // instantiated here: one more copy of the horizon loop per model for the one wave that owns row 0.)
    const int step_mode = (__any(orow != -2) || a.states != nullptr) ? 2 : 0;
    // the nu MFMAs of support point s (B straight from the AGPRs, or from the four-vectors `b` read out of LDS)
    auto mac_s = [&](auto sc, float w, const kf32x4_t (&b)[P4], kf32x4_t (&D)[NU]) {
      constexpr int s = decltype(sc)::value;
#pragma unroll
      for (int q = 0; q < P4; ++q) {
        const int i0 = s * NU + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if defined(MPPI_KMPPI_EXP) && (MPPI_KMPPI_EXP & 2)   // experiment: no matrix instructions (one per support point keeps the operands alive)
          if (q == 0 && i == 0) D[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, i0 < NA ? tha[i0 < NA ? i0 : 0] : b[q][i], D[0], 0, 0, 0);
          continue;
#endif
          if (i0 < NA) D[4 * q + i] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, tha[i0 + i < NA ? i0 + i : 0], D[4 * q + i], 0, 0, 0);   // :665
          else D[4 * q + i] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, b[q][i], D[4 * q + i], 0, 0, 0);
        }
      }
    };
    auto lds_s = [&](auto sc, kf32x4_t (&b)[P4]) {
      constexpr int s = decltype(sc)::value;
#pragma unroll
      for (int q = 0; q < P4; ++q) {
        const int i0 = s * NU + 4 * q;
        if (i0 >= NA) b[q] = thx_me[((i0 - NA) >> 2) * K1_BLOCK];
      }
    };
    // all S4 == SMAX support points of one tile: MFMAs only (operator row of the NEXT group of four and the
    // LDS-resident control points of the NEXT support point are read one step ahead -- one wave per SIMD,
    // nobody else hides an LDS round trip)
    auto macs_full = [&](int t0, kf32x4_t (&D)[NU]) {
#pragma unroll
      for (int n = 0; n < NU; ++n) D[n] = kf32x4_t{0.f, 0.f, 0.f, 0.f};
      const kf32x4_t* Wr = reinterpret_cast<const kf32x4_t*>(Wl + (t0 + (lane & 3)) * SPW);
      kf32x4_t w = Wr[0], wn = w;
      kf32x4_t b[P4], bn[P4];
#pragma unroll
      for (int q = 0; q < P4; ++q) b[q] = bn[q] = kf32x4_t{0.f, 0.f, 0.f, 0.f};
      lds_s(std::integral_constant<int, 0>{}, b);
      static_for<0, SMAX>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s % 4 == 0 && s + 4 < SMAX) wn = Wr[s / 4 + 1];
        if constexpr (s + 1 < SMAX) lds_s(std::integral_constant<int, s + 1>{}, bn);
        __builtin_amdgcn_sched_barrier(0);
        mac_s(sc, w[s % 4], b, D);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < P4; ++q) b[q] = bn[q];
        if constexpr (s % 4 == 3) w = wn;
      });
    };
    // the whole horizon, for one wave-uniform step mode (decided once: no branch between a tile's MFMAs and its steps)
    auto horizon = [&](auto mode_c) {
      constexpr int MODE = decltype(mode_c)::value;
      if (S4 == SMAX) {
        for (int t0 = 0; t0 < Thor; t0 += 4) {
          kf32x4_t D[NU];
          macs_full(t0, D);
          kmppi_steps4<Model, MODE>(a, ac, model, tb, k, active, orow, t0, D, x, vprev, rollout, pert);
        }
      } else {
        for (int t0 = 0; t0 < Thor; t0 += 4) {
          kf32x4_t D[NU];
  #pragma unroll
          for (int n = 0; n < NU; ++n) D[n] = kf32x4_t{0.f, 0.f, 0.f, 0.f};
          const kf32x4_t* Wr = reinterpret_cast<const kf32x4_t*>(Wl + (t0 + (lane & 3)) * SPW);
          static_for<0, SMAX / 4>([&](auto gc) {
            constexpr int s4 = decltype(gc)::value;
            if (4 * s4 < S) {                             // wave-uniform; beyond S: theta' = 0, W = 0
              const kf32x4_t w = Wr[s4];
              static_for<0, 4>([&](auto jc) {
                constexpr int s = 4 * s4 + decltype(jc)::value;
                kf32x4_t b[P4];
  #pragma unroll
                for (int q = 0; q < P4; ++q) b[q] = kf32x4_t{0.f, 0.f, 0.f, 0.f};
                lds_s(std::integral_constant<int, s>{}, b);
                mac_s(std::integral_constant<int, s>{}, w[s % 4], b, D);
              });
            }
          });
          kmppi_steps4<Model, MODE>(a, ac, model, tb, k, active, orow, t0, D, x, vprev, rollout, pert);
        }
      }
    };
    if (step_mode == 0) horizon(std::integral_constant<int, 0>{});
    else horizon(std::integral_constant<int, 2>{});
    if (a.use_terminal) rollout += model.terminal(x);                    // :324-328
    const float total = rollout + pert;                                  // :416
    if (active) {
      a.cost[k] = total;
      if (a.pert != nullptr) a.pert[k] = pert;
    }
    if (!a.kw) {
      const float bm = wave_min<float>(active ? total : inf_v<float>());   // one minimum per 64 samples
      if (lane == 0 && kraw < a.K) a.block_min[kraw / WAVE] = bm;
    } else {
      // ---- this workgroup's part of the theta update (mppi.py:254-259, :679), as the on-chip MPPI command does it for U:
      // weights relative to the WORKGROUP's minimum, eta_b, P_b[i] = sum_k w_k (theta'_k[i] - theta[i]) over its 256 samples.
      // The lane still holds its bounded control points (AGPRs + LDS): nothing is re-read or generated again -- the
      // stand-alone K3 re-created all S*nu rows per sample (18 us + a launch at C3-sized work).
      const float inv_lambda = 1.f / a.lambda_;
      const float beta_b = block_min<float>(active ? total : inf_v<float>(), red);
      const float wk = active ? weight_of<float>(total, beta_b, inv_lambda) : 0.f;
      const float eta_b = block_sum<float>(wk, red);
      const bool live = __ballot(wk != 0.f) != 0ull;      // a wave of exactly-zero weights adds exactly nothing
      const int wv = threadIdx.x / WAVE;
      const int ncol = S * NU;
      static_for<0, NWT>([&](auto tc) {
        constexpr int TI = decltype(tc)::value;
        __builtin_amdgcn_sched_barrier(0);     // one tile at a time: six tiles' operands in flight at once do not fit the file
        if (TI * 64 < ncol) {
          float acc[64];
          if (live) {
#pragma unroll
            for (int q4 = 0; q4 < 16; ++q4) {
              const int i0 = TI * 64 + 4 * q4;             // (compile-time: TI and q4 are)
              kf32x4_t v;
              if (TI * 64 + 4 * q4 < NA) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = from_agpr(tha[(TI * 64 + 4 * q4 + c) < NA ? (TI * 64 + 4 * q4 + c) : 0]);
              } else if (TI * 64 + 4 * q4 < SMAX * NU) {
                v = thx_me[((TI * 64 + 4 * q4 - NA) >> 2) * K1_BLOCK];
              } else {
                v = kf32x4_t{0.f, 0.f, 0.f, 0.f};
              }
              // (columns beyond S*nu hold leftovers of the padding: every column is reduced on its own and those are never
              // stored -- no select, whose 384 wave-uniform conditions cost 700 spilled SGPRs)
              const kf32x4_t t4 = *reinterpret_cast<const kf32x4_t*>(th0 + i0);
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[4 * q4 + c] = wk * (v[c] - t4[c]);
            }
            exw[(wv * NWT + TI) * 64 + lane] = wave_reduce_transpose64<float>(acc);
          } else {
            exw[(wv * NWT + TI) * 64 + lane] = 0.f;
          }
        }
      });
      __syncthreads();
      for (int i = threadIdx.x; i < ncol; i += K1_BLOCK)      // one combine over the four waves, in wave order
        a.P_part[(long long)chunk * a.kw_jpad + i] = (exw[i] + exw[NWT * 64 + i]) + (exw[2 * NWT * 64 + i] + exw[3 * NWT * 64 + i]);
      if (threadIdx.x == 0) {
        a.eta_part[chunk] = eta_b;
        a.block_min[chunk] = beta_b;
      }
    }
  }
  if (a.tstamp != nullptr) {
    __syncthreads();
    stamp_exit(a.tstamp);
  }
}

// returns MPPI_E_UNSUPPORTED where the fused form does not exist (caller: two-launch form)
template <class Model, typename T>
static int launch_rollout_kmppi(const KArgs<T>& a, hipStream_t st) {
  constexpr int NU = Model::NU;
  if constexpr (!std::is_same<T, float>::value || !KmppiFuse<NU>::OK || !KmppiModelOk<Model>::value) {
    return MPPI_E_UNSUPPORTED;
  } else {
    if (a.S <= 0 || a.S > KmppiFuse<NU>::SMAX || a.theta == nullptr || a.W == nullptr) return MPPI_E_UNSUPPORTED;
    if (!a.diag || a.coloured || a.B != nullptr || a.smooth_w != 0.f || a.M != 1 || a.n_env != 1) return MPPI_E_UNSUPPORTED;
    if (a.noise_src != MPPI_NOISE_TNK4 && a.noise_src != MPPI_NOISE_PHILOX) return MPPI_E_UNSUPPORTED;
    // rng="philox7": this kernel's in-lane generator is the ten-round one (a run-time round count cost it 9 us of 72: 96 rows per
    // sample generated in its prologue); seven-round problems come with their rows in memory (the host runs the generator launch)
    if (a.seven && a.noise_src == MPPI_NOISE_PHILOX) return MPPI_E_UNSUPPORTED;
    const int S4 = (a.S + 3) & ~3, T4 = (a.Tn + 3) & ~3;
    const int in_lds = S4 * NU > KmppiRegs<NU>::NA ? S4 * NU - KmppiRegs<NU>::NA : 0;     // control points per lane beyond the AGPRs
    const size_t smem0 = (size_t)(2 * ((a.J + 3) & ~3) + ((a.S * NU + 3) & ~3) + T4 * (S4 + 4) + in_lds * K1_BLOCK) * sizeof(float);
    if (smem0 > 160 * 1024) return MPPI_E_UNSUPPORTED;
    // the theta update inside the launch (a.kw asked for by mppi_command_kmppi): needs theta, the reduction scratch and the four
    // waves' column sums in LDS as well; where that does not fit the kernel leaves the per-wave minima for K3 as before
    constexpr int NWT = (KmppiFuse<NU>::SMAX * NU + 63) / 64;
    const size_t smem_w = smem0 + (size_t)(NWT * 64 + 4 + 4 * NWT * 64) * sizeof(float);
    const bool kw = a.kw != 0 && smem_w <= 160 * 1024 && a.n_sampler == 0 && (a.K + K1_BLOCK - 1) / K1_BLOCK <= 8192;
    const size_t smem = kw ? smem_w : smem0;
    const int nchunks = (a.K + K1_BLOCK - 1) / K1_BLOCK;
    static const int n_cu = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return n > 0 ? n : 256;
    }();
    const dim3 grid(nchunks < n_cu ? nchunks : n_cu), block(K1_BLOCK);
    KArgs<float> b = a;
    b.kw = kw ? 1 : 0;
    if (kw) onchip_carve(b);               // one partial record per 256 samples: block_min / eta_part / P_part[.][kw_jpad]
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    profile_next_events(&ev0, &ev1, &b.tstamp);
#define MPPI_LAUNCHK(KERNEL)                                                                                 \
  do {                                                                                                       \
    if (smem > 64 * 1024)                                                                                    \
      (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (ev1 != nullptr) hipExtLaunchKernelGGL(KERNEL, grid, block, smem, st, ev0, ev1, 0, b);                \
    else hipLaunchKernelGGL(KERNEL, grid, block, smem, st, b);                                               \
  } while (0)
    if (a.noise_src == MPPI_NOISE_PHILOX) MPPI_LAUNCHK((rollout_kmppi_kernel<Model, MPPI_NOISE_PHILOX>));
    else MPPI_LAUNCHK((rollout_kmppi_kernel<Model, MPPI_NOISE_TNK4>));
#undef MPPI_LAUNCHK
    const int e = (int)hipGetLastError();
    return e != 0 ? e : (kw ? MPPI_OK_KMPPI_W : 0);
  }
}

}  // namespace mppi
