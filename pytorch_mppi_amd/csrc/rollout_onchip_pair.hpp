// rollout_onchip_pair.hpp -- the on-chip command (rollout_onchip.hpp) with TWO waves per SIMD (round 6).
//
// The on-chip kernel runs one wave per SIMD: the sample's bounded noise waits in all 512 registers of the lane, and K = 65536 is one
// wave per SIMD of samples anyway.  With a single wave nothing hides a wait: VALU active 0.55 of the wave's cycles, waiting 0.24
// (profiles/pmc_onchip_valu.json).  Measured on the product kernel with the weighting and the keeping knocked out
// (profiles/r06_e_onchip_two_waves.txt): generate + roll out takes 44.9 us at one wave per SIMD and 34.4 us per 65536 samples when two
// workgroups share a CU.  Two workgroups cannot share a CU with anything kept (registers, LDS) -- but two WAVES can share a sample:
//
//   * a workgroup is 8 waves for the same 256 samples; waves p and p + 4 (one SIMD) are a PAIR, lane l of both is sample 64 p + l;
//   * the horizon's super-steps go to the two waves in alternating chunks of CH (= the rows generated together): wave A (h = 0)
//     owns chunks 0, 2, 4 ..., wave B (h = 1) chunks 1, 3, 5 ...  Each wave generates, bounds and KEEPS the rows of its chunks -- the
//     first KR local super-steps in its registers, then LDS, then the spill array -- and later forms the column sums of exactly those
//     rows: generation, keeping and the weighting phase are independent per (sample, row) and split without any exchange;
//   * the rollout is sequential in t.  The state (and the running cost and the action cost accumulated so far, so that every sum is
//     taken in the order the one-wave kernel takes it) goes from wave to wave through LDS at every chunk boundary, behind a progress
//     counter per pair (release / acquire on LDS): a wave generates its chunk, waits until its partner has rolled out the chunk in
//     front of it -- which the partner did while this wave was generating, so the wait is rarely one --, takes over, rolls its CH
//     timesteps, hands over, keeps.  No workgroup barrier between the prologue and the weighting phase: pairs drift freely.  (A first
//     version with two barriers per iteration was SLOWER than the one-wave kernel, 80.4 against 76.7 us: between the barriers one
//     wave of the pair rolled out alone, with nothing to hide its latencies -- profiles/r06_f_onchip_pair_check.txt.)
//     While one wave of the pair waits -- LDS tables, the hand-over, spill stores, the reductions' cross-lane moves -- the other
//     issues its Philox multiplies.
//
// Results: bit for bit those of rollout_onchip_kernel (same Philox counters, same arithmetic per row, same order of every sum; the
// column sums' lane tree does not depend on where a column sits in its tile) -- tests/test_gpu_onchip.py compares the two forms.
// Scope: what the launcher below accepts -- MPPI and SMPPI (|noise| cost, u_scale, base sequence, 1/dt rescaling and the smoothness cost,
// whose operand -- the previous timestep's action -- rides in the hand-over too), diagonal Sigma, a spill array, the models of
// onchip_pair_model_ok; everything else stays on the one-wave kernel.
#pragma once
// (included from rollout.hpp behind rollout_onchip.hpp)

namespace mppi {

template <int NU, bool PLAIN = true>
struct OnChipPair {
  using OC = OnChip<NU>;
  static constexpr int P4 = OC::P4, TT = OC::TT, SW = OC::SW, TRW = OC::TRW, TC = OC::TC;
  // super-steps per chunk: generated together, owned by one wave.  With the SMPPI terms compiled in, one fewer: their operands
  // cost the generator's live set the registers of a super-step's z and v (the four-super-step form spilled 14-40 VGPRs)
  static constexpr int CH = (PLAIN || OC::PB < 2) ? OC::PB : OC::PB - 1;
  static constexpr int KT = MPPI_PAIR_KT, KR = KT * SW;          // weighting tiles / local super-steps a wave keeps in registers
  static constexpr int NCA = (KR + CH - 1) / CH;                 // local chunks that touch the registers (static indices)
  static constexpr bool OK = OC::OK;
};

template <int NU>
struct OnChipRowG {
  float g[Stream<NU>::P4 * 4];
  __device__ __forceinline__ void load(const float* table, int ss) {       // one super-step's row of a per-(t,n) LDS table
    constexpr int P4 = Stream<NU>::P4;
    const float4* __restrict__ pg = reinterpret_cast<const float4*>(__builtin_assume_aligned(table, 16)) + ss * P4;
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      const float4 g4 = pg[i];
      g[4 * i] = g4.x; g[4 * i + 1] = g4.y; g[4 * i + 2] = g4.z; g[4 * i + 3] = g4.w;
    }
  }
};

// PLAIN: no |noise| cost, u_scale == 1, no SMPPI terms (the common case, compiled without them); PLAIN = false: those terms as the
// one-wave kernel applies them, and the previous timestep's action (the smoothness cost's operand) in the hand-over
template <class Model, bool SEVEN = false, bool PLAIN = true>
__global__ void __launch_bounds__(2 * K1_BLOCK) rollout_onchip_pair_kernel(const KArgs<float> a, const int nsl, const int nsm) {
  using T = float;
  constexpr int NX = Model::NX, NU = Model::NU;
  using OP = OnChipPair<NU, PLAIN>;
  constexpr int P4 = OP::P4, TT = OP::TT, CH = OP::CH, SW = OP::SW, TRW = OP::TRW, TC = OP::TC, KT = OP::KT, KR = OP::KR, NCA = OP::NCA;
  constexpr int NT = 2 * K1_BLOCK;
  static_assert(K1_BLOCK == 256 && K1_BLOCK == BLOCK, "a pair kernel workgroup is eight waves over one 256-sample record");
  stamp_entry(a.tstamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nss = (a.Tn + TT - 1) / TT;
  const int nch = (nss + CH - 1) / CH, nit = (nch + 1) / 2, nls = nit * CH, ntl = (nls + SW - 1) / SW;
  const int Jp = nss * P4 * 4;
  T* Ue = reinterpret_cast<T*>(smem_raw);
  T* Um = Ue + Jp;
  T* G = Um + Jp;
  T* red = G + Jp;                       // [8] | the pairs' progress counters [4] (+ 4 unused)
  int* flag = reinterpret_cast<int*>(red + 8) + ((threadIdx.x >> 6) & 3);   // this pair's: chunks rolled out so far
  T* sh = red + 16;                      // the hand-over [NX + 2][256], later the waves' column sums [8][ntl][64]
  constexpr int HAND = NX + 2 + (PLAIN ? 0 : NU);
  const int hand_n = HAND * K1_BLOCK, ex_n = 8 * ntl * 64;
  float4* keepL = reinterpret_cast<float4*>(sh + (hand_n > ex_n ? hand_n : ex_n));
  for (int j = threadIdx.x; j < Jp; j += NT) {
    const bool in = j < a.J;
    const int n = j % NU;
    const T u = in ? u_base(a, j) : T(0);
    Ue[j] = u;
    Um[j] = in ? u + a.mu[n] : T(0);
    G[j] = in ? a.lambda_ * ((a.B != nullptr ? u_eff(a, j) : u) * a.sinv[n * NU + n]) : T(0);   // always the true U
  }
  const int lane = threadIdx.x & (WAVE - 1), wv8 = threadIdx.x / WAVE;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);      // 0: even chunks, 1: odd chunks (wave-uniform)
  const int s256 = threadIdx.x & (K1_BLOCK - 1);                        // the sample's slot in the workgroup
  const int kraw = blockIdx.x * K1_BLOCK + s256;
  const bool active = kraw < a.K;
  const int k = active ? kraw : a.K - 1;
  const int orow = overwrite_row(a, a.k_offset + k);
  const Model model(a);
  T x[NX];
  if (h == 0) {
    // the state starts its way in the hand-over like every later one: no register holds it across a barrier
    const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
    for (int i = 0; i < NX; ++i) sh[i * K1_BLOCK + s256] = s0[i];                                 // mppi.py:302-305
    sh[NX * K1_BLOCK + s256] = T(0);
    sh[(NX + 1) * K1_BLOCK + s256] = T(0);
    if constexpr (!PLAIN) {
#pragma unroll
      for (int n = 0; n < NU; ++n) sh[(NX + 2 + n) * K1_BLOCK + s256] = T(0);
    }
  }
  if (threadIdx.x < 4) reinterpret_cast<int*>(red + 8)[threadIdx.x] = 0;
  ActionConsts<T, NU> ac;
  ac.load(a, nullptr);
  T keepR[KR * P4 * 4];
#pragma unroll
  for (int i = 0; i < KR * P4 * 4; ++i) keepR[i] = T(0);              // local super-steps this wave never owns read as zero noise
  __syncthreads();
  const StepTables<T> tb{Ue, Um, G, nullptr, kraw - lane};
  const long long kg = a.k_offset + k;
  const int M0 = KR + nsl, M1 = M0 + nsm;                               // local super-steps [M0, M1) wait in memory
  // [local row][workgroup][thread][4]: a wave-uniform base per row + a 32-bit lane offset (the per-lane 64-bit pointer of the
  // one-wave kernel had its row addresses hoisted out of the loops and spilled: six VGPR pairs in scratch)
  char* __restrict__ spill_blk = reinterpret_cast<char*>(a.spill) + (size_t)blockIdx.x * NT * 16;
  const size_t Kp16 = (size_t)gridDim.x * NT * 16;
  const unsigned lane_off = threadIdx.x * 16u;
  auto spill_at = [&](int row) __attribute__((always_inline)) {
    asm volatile("" : "+s"(row));           // (computed where it is used: scalar multiply + add, no address held in VGPRs across the loops)
    return reinterpret_cast<float4*>(spill_blk + (size_t)row * Kp16 + lane_off);
  };
  const bool null_in_wave = __any(orow == -1);
  T vprev[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) vprev[n] = T(0);
  T rollout, pert;

  for (int it = 0; it < nit; ++it) {
    const int c = 2 * it + h, ss0 = c * CH;
    const bool own = ss0 < nss;                                          // (the odd wave may have one chunk fewer)
    T zb[CH][P4 * 4], vb[CH][P4 * 4];
    if (own) {
#pragma unroll
      for (int b = 0; b < CH; ++b) {
#pragma unroll
        for (int i = 0; i < P4; ++i) {
          T r[4];
          philox_normal4<T, SEVEN ? 7 : 10>(a.seed, a.call, kg, (long long)(ss0 + b) * P4 + i, r);   // rows past the horizon: unused
          zb[b][4 * i + 0] = r[0]; zb[b][4 * i + 1] = r[1]; zb[b][4 * i + 2] = r[2]; zb[b][4 * i + 3] = r[3];
        }
      }
#pragma unroll
      for (int b = 0; b < CH; ++b) {
        const int ss = ss0 + b;
        OnChipRow<NU> row;
        row.load(tb, ss < nss ? ss : nss - 1, false);
        onchip_actions<NU, true>(ac, row, orow, null_in_wave, zb[b], vb[b], !PLAIN);
      }
    }
    // (Holding only v and forming eps' = v - U again where it is used -- 48 registers fewer in the generator's live set -- was tried:
    //  generate + roll out + keep went from 44.5 to 50.7 us, profiles/r06_g_onchip_pair_check.txt)
    auto keep_dyn = [&](int b) {
      const int ls = it * CH + b;                                        // local super-step
      if (ls >= KR && ls < M0) {
        asm volatile("; keep in LDS");
#pragma unroll
        for (int i = 0; i < P4; ++i)
          keepL[((ls - KR) * P4 + i) * NT + threadIdx.x] = make_float4(zb[b][4 * i], zb[b][4 * i + 1], zb[b][4 * i + 2], zb[b][4 * i + 3]);
        asm volatile("; kept in LDS" ::: "memory");
      } else if (ls >= M0 && ls < M1) {
        asm volatile("; keep in memory");
#pragma unroll
        for (int i = 0; i < P4; ++i)
          *spill_at((ls - M0) * P4 + i) = make_float4(zb[b][4 * i], zb[b][4 * i + 1], zb[b][4 * i + 2], zb[b][4 * i + 3]);
        asm volatile("; kept in memory" ::: "memory");
      }
    };
    if (own) {
      // ---- wait for the partner's chunk c - 1, take over, roll this chunk out, hand over ----
      // (no workgroup barrier: the pairs run free of each other, and a wave only ever waits here -- for a rollout of CH timesteps that
      //  its partner started while this wave was still generating)
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < c) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = sh[i * K1_BLOCK + s256];
      rollout = sh[NX * K1_BLOCK + s256];
      pert = sh[(NX + 1) * K1_BLOCK + s256];
      if constexpr (!PLAIN) {
#pragma unroll
        for (int n = 0; n < NU; ++n) vprev[n] = sh[(NX + 2 + n) * K1_BLOCK + s256];
      }
#pragma unroll
      for (int b = 0; b < CH; ++b) {
        const int ss = ss0 + b;
        OnChipRowG<NU> rg;
        rg.load(tb.G, ss < nss ? ss : nss - 1);
        OnChipRow<NU> row;
#pragma unroll
        for (int q = 0; q < P4 * 4; ++q) row.g[q] = rg.g[q];
        onchip_steps<Model>(a, ac, model, row, ss, zb[b], vb[b], x, vprev, rollout, pert, PLAIN);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) sh[i * K1_BLOCK + s256] = x[i];
      sh[NX * K1_BLOCK + s256] = rollout;
      sh[(NX + 1) * K1_BLOCK + s256] = pert;
      if constexpr (!PLAIN) {
#pragma unroll
        for (int n = 0; n < NU; ++n) sh[(NX + 2 + n) * K1_BLOCK + s256] = vprev[n];
      }
      __hip_atomic_store(flag, c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#if defined(MPPI_PAIR_EXP) && (MPPI_PAIR_EXP & 4)        // experiment (tools/micro/onchip_pair_check.hip): nothing kept
      continue;
#endif
      // ---- keep eps' of this chunk: registers (static index -> a chain of uniform compares over the iteration), LDS, memory ----
      bool done = false;
      static_for<0, NCA>([&](auto cc) {
        constexpr int C = decltype(cc)::value;
        if (it == C) {
          static_for<0, CH>([&](auto bb) {
            constexpr int B = decltype(bb)::value, LS = C * CH + B;
            if constexpr (LS < KR) {
#pragma unroll
              for (int q = 0; q < P4 * 4; ++q) keepR[LS * P4 * 4 + q] = zb[B][q];
            } else {
              keep_dyn(B);
            }
          });
          done = true;
        }
      });
      if (!done) {
#pragma unroll
        for (int b = 0; b < CH; ++b) keep_dyn(b);
      }
    }
  }
  // local tiles: [0, tA) registers | [tA, tB) LDS | [tB, tC) memory | tile tC: its first `mix` super-steps memory, the rest -- and
  // every later tile -- generated a second time (nsm need not be whole tiles: the split that balances fetch and VALU is not)
  const int tA = ntl < KT ? ntl : KT;
  const int tB = tA + nsl / SW < ntl ? tA + nsl / SW : ntl;
  const int tC = tB + nsm / SW < ntl ? tB + nsm / SW : ntl;
  const int mix = tC < ntl ? nsm % SW : 0;
  const int tM = tC + (mix > 0 ? 1 : 0);                                 // tiles [tB, tM) read rows that waited in memory
  float4 pf[TRW];
#if defined(MPPI_PAIR_EXP) && (MPPI_PAIR_EXP & 8)        // experiment: the tiles that waited in memory are not fetched (nor summed)
  int mt = tM;
#else
  int mt = tB;
#endif

  // rows of this wave that exist: its local super-steps map to increasing super-steps of the horizon, so they are a prefix
  int nreal = 0;
  for (int ls = 0; ls < nls; ++ls) nreal += ((2 * (ls / CH) + h) * CH + ls % CH) < nss ? P4 : 0;
  const int nstored = nreal < M1 * P4 ? nreal : M1 * P4;                 // local rows [M0 P4, nstored) are in the array
  auto fetch = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TRW; ++i) {
      // (a row never stored is not fetched: the last stored row is read again -- a hit -- and not used below.  No branch: the rows
      //  of a tile are one batch of loads)
      const int row = tile * TRW + i < nstored ? tile * TRW + i : nstored - 1;
      pf[i] = *spill_at(row - tB * TRW);
    }
  };
  // the first tile that waited in memory takes off now: it flies while the partner finishes the last chunk and the workgroup
  // agrees on its minimum (whether this wave's weights are all zero is not known yet: a dead wave fetches this one tile in vain)
  if (mt < tM && nstored > tB * TRW) fetch(mt);
  // every chunk rolled out: the last hand-over holds the final state and the two cost sums
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < nch) __builtin_amdgcn_s_sleep(1);
  {
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = sh[i * K1_BLOCK + s256];
    rollout = sh[NX * K1_BLOCK + s256];
    pert = sh[(NX + 1) * K1_BLOCK + s256];
  }
  if (a.use_terminal) rollout += model.terminal(x);
  const T total = rollout + pert;
  if (active && h == 0) {
    a.cost[k] = total;
    if (a.pert != nullptr) a.pert[k] = pert;
  }

#if defined(MPPI_PAIR_EXP) && (MPPI_PAIR_EXP & 1)        // experiment: no weighting phase
  return;
#endif
  // ---- W: every wave forms the column sums of ITS rows; the workgroup's record as in rollout_onchip_kernel ----
  T* ex = sh;                                                            // (the hand-over is dead behind block_min's barriers)
  const T inv_lambda = T(1) / a.lambda_;
  const T beta_b = block_min<T>(active ? total : inf_v<T>(), red);       // (the first four waves' entries: the 256 samples once)
  const T wk = active ? weight_of<T>(total, beta_b, inv_lambda) : T(0);
  const T eta_b = block_sum<T>(wk, red);
  const bool live = __ballot(wk != T(0)) != 0ull;
  auto column_sums = [&](int tile, const T (&e)[TRW][4]) __attribute__((always_inline)) {
    T acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = i < TC ? wk * e[(i < TC ? i : 0) / 4][i % 4] : T(0);
    ex[(wv8 * ntl + tile) * 64 + lane] = wave_reduce_transpose64<float>(acc);
  };
  auto from_registers = [&](int tile) __attribute__((always_inline)) {
    T e[TRW][4];
    static_for<0, KT>([&](auto tt) {
      constexpr int TI = decltype(tt)::value;
      if (tile == TI) {
        asm volatile("; register tile %0" ::"n"(TI));
#pragma unroll
        for (int i = 0; i < TRW; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) e[i][q] = keepR[(TI * TRW + i) * 4 + q];
      }
    });
    column_sums(tile, e);
  };
  if (!live) {
    for (int tile = 0; tile < ntl; ++tile) ex[(wv8 * ntl + tile) * 64 + lane] = T(0);
  } else {
    for (int tile = 0; tile < tA; ++tile) from_registers(tile);
    for (int tile = tA; tile < tB; ++tile) {
      T e[TRW][4];
#pragma unroll
      for (int i = 0; i < TRW; ++i) {
        const float4 q4 = keepL[((tile - tA) * TRW + i) * NT + threadIdx.x];
        e[i][0] = q4.x; e[i][1] = q4.y; e[i][2] = q4.z; e[i][3] = q4.w;
      }
      column_sums(tile, e);
    }
    // The remaining tiles: rows that waited in memory and rows generated a second time.  The weighting phase of this kernel is
    // bound either by the fetch of the memory rows (~8 TB/s of them out of the Infinity Cache, chip-wide) or by its VALU work (the
    // column sums and the second generation, two waves per SIMD); the two proceed side by side, a memory tile's successor in
    // flight while a regenerated tile is computed, and the launcher picks the split that makes the larger of them smallest
    // (onchip_pair_geometry; profiles/r06_h_onchip_pair_check.txt, r06_k_onchip_pair_check.txt: everything fetched 67.1 us, the last two
    // tiles generated again 60.4, splits inside a tile 62.3-62.4).
    // (The one-wave kernel found all-fetch best: there a second generation runs at one wave per SIMD.)
    int kd = blockIdx.x * K1_BLOCK + s256;
    asm volatile("" : "+v"(kd));
    const long long kgd = a.k_offset + (kd < a.K ? kd : a.K - 1);
    int dt = tM;                                                         // the next tile generated again in full
    for (int step = 0; mt < tM || dt < ntl; ++step) {
      // one body for every kind of tile: its first m super-steps out of `pf`, the rest generated again
      const bool from_memory = mt < tM && ((step & 1) == 0 || dt >= ntl);
      const int tile = from_memory ? mt : dt;
      const int m = from_memory ? (mt < tC ? SW : mix) : 0;
      T acc[64];                                                         // (the products go straight into the reduction's operands)
#pragma unroll
      for (int i = TC; i < 64; ++i) acc[i] = T(0);
#pragma unroll
      for (int g = 0; g < SW; ++g) {
        const int ls = tile * SW + g;
        const int ss = (2 * (ls / CH) + h) * CH + ls % CH;
        if (g < m) {
#pragma unroll
          for (int i = 0; i < P4; ++i) {
            const float4 q4 = pf[g * P4 + i];
            const bool real = ls * P4 + i < nstored;                     // (rows past the horizon were never stored)
            acc[(g * P4 + i) * 4 + 0] = real ? wk * q4.x : T(0);
            acc[(g * P4 + i) * 4 + 1] = real ? wk * q4.y : T(0);
            acc[(g * P4 + i) * 4 + 2] = real ? wk * q4.z : T(0);
            acc[(g * P4 + i) * 4 + 3] = real ? wk * q4.w : T(0);
          }
        } else if (ss < nss) {
          T zg[P4 * 4], vg[P4 * 4];
#pragma unroll
          for (int i = 0; i < P4; ++i) {
            T r[4];
            philox_normal4<T, SEVEN ? 7 : 10>(a.seed, a.call, kgd, (long long)ss * P4 + i, r);
            zg[4 * i + 0] = r[0]; zg[4 * i + 1] = r[1]; zg[4 * i + 2] = r[2]; zg[4 * i + 3] = r[3];
          }
          OnChipRow<NU> row;
          row.load(tb, ss, false);
          onchip_actions<NU, true>(ac, row, orow, null_in_wave, zg, vg, !PLAIN);
#pragma unroll
          for (int q = 0; q < P4 * 4; ++q) acc[g * P4 * 4 + q] = wk * zg[q];
        } else {
#pragma unroll
          for (int q = 0; q < P4 * 4; ++q) acc[g * P4 * 4 + q] = T(0);
        }
      }
      if (from_memory) {
        if (++mt < tM) fetch(mt);                                        // `pf` is consumed: the next memory tile takes off
      } else {
        ++dt;
      }
      ex[(wv8 * ntl + tile) * 64 + lane] = wave_reduce_transpose64<float>(acc);
    }
  }
  __syncthreads();
  // one combine over the four waves that own a column, in wave order
  for (int j = threadIdx.x; j < a.Jpad; j += NT) {
    const int r = j >> 2, ss = r / P4;
    T sum = T(0);
    if (ss < nch * CH) {
      const int cidx = ss / CH, hh = cidx & 1, ls = (cidx >> 1) * CH + ss % CH;
      const int lr = ls * P4 + r % P4, lt = lr / TRW, o = lt * 64 + (lr % TRW) * 4 + (j & 3);
      const T* e0 = ex + (long long)(hh * 4) * ntl * 64 + o;
      sum = (e0[0] + e0[ntl * 64]) + (e0[2 * ntl * 64] + e0[3 * ntl * 64]);
    }
    a.P_part[(long long)blockIdx.x * a.Jpad + j] = sum;
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

// Which models take the pair kernel: those whose instantiation fits the 256 registers of a wave without scratch and was measured
// faster than the one-wave kernel (tools/micro/onchip_pair_check.hip; tests/test_kernel_resources.py reads the code objects).
// LinearGoal (12, 4) -- 64 kept values per tile, 16-row tiles -- spills 43 VGPRs: it stays on the one-wave kernel.
template <class Model>
struct onchip_pair_model_ok : std::false_type {};
template <>
struct onchip_pair_model_ok<IntegratorModel<float, 16, 12>> : std::true_type {};
#ifdef MPPI_PAIR_ANY_MODEL   // (the check tool: any model)
#define MPPI_PAIR_MODEL_OK(Model) true
#else
#define MPPI_PAIR_MODEL_OK(Model) onchip_pair_model_ok<Model>::value
#endif

// 0: never | 1: wherever it applies (the default once measured) -- MPPI_ONCHIP_PAIR
// (read at every launch, not once: an A/B switch that tests flip inside one process)
static inline int onchip_pair_mode() {
  const char* e = getenv("MPPI_ONCHIP_PAIR");
  return e ? atoi(e) : MPPI_ONCHIP_PAIR_DEFAULT;
}

// returns MPPI_OK_ONCHIP when the launch was issued, a positive HIP error, or -1: not this form (the one-wave kernel takes it)
template <class Model>
static int launch_rollout_onchip_pair(const KArgs<float>& a, hipStream_t st) {
  if constexpr (!MPPI_PAIR_MODEL_OK(Model)) {
    return -1;
  } else {
  constexpr int NU = Model::NU, NX = Model::NX;
  using OP = OnChipPair<NU, true>;
  if (a.spill == nullptr || a.diag == 0) return -1;
  const bool plain = !a.abs_cost && a.u_scale == 1.f && a.e_scale == 1.f && a.smooth_w == 0.f && a.B == nullptr;
  const OnChipPairGeometry g = onchip_pair_geometry(NU, NX, a.Tn, plain);
  if (!g.ok || g.P4 != OP::P4 || g.TT != OP::TT || g.SW != OP::SW || g.KR != OP::KR ||
      g.CH != (plain ? OnChipPair<NU, true>::CH : OnChipPair<NU, false>::CH)) return -1;
  if (g.nch < 4) return -1;                                            // a horizon of fewer than four chunks: nothing to alternate
  if ((long long)g.nsm * g.P4 * a.nkc * 2 * K1_BLOCK * 4 > a.spill_cap) return -1;     // the caller's array is the one-wave form's size
  const dim3 grid(a.nkc), block(2 * K1_BLOCK);
  KArgs<float> b = a;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  profile_next_events(&ev0, &ev1, &b.tstamp);
  const size_t smem = g.smem;
#define MPPI_PAIR_LAUNCH(KERNEL)                                                                                  \
  do {                                                                                                            \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (ev1 != nullptr) hipExtLaunchKernelGGL(KERNEL, grid, block, smem, st, ev0, ev1, 0, b, g.nsl, g.nsm);       \
    else hipLaunchKernelGGL(KERNEL, grid, block, smem, st, b, g.nsl, g.nsm);                                      \
  } while (0)
  if (a.seven) {
    if constexpr (onchip_seven_ok<Model>::value) {
      if (plain) MPPI_PAIR_LAUNCH((rollout_onchip_pair_kernel<Model, true, true>));
      else MPPI_PAIR_LAUNCH((rollout_onchip_pair_kernel<Model, true, false>));
    } else return -1;
  } else if (plain) MPPI_PAIR_LAUNCH((rollout_onchip_pair_kernel<Model, false, true>));
  else MPPI_PAIR_LAUNCH((rollout_onchip_pair_kernel<Model, false, false>));
#undef MPPI_PAIR_LAUNCH
  const int e = (int)hipGetLastError();
  if (e == 0 && onchip_pair_launched) onchip_pair_launched();
  return e != 0 ? e : MPPI_OK_ONCHIP;
  }
}

}  // namespace mppi
