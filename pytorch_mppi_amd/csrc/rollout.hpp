// rollout.hpp -- K1: the fused noise -> bound -> action-cost -> T-step rollout -> running-cost
// kernel.  One lane = one sample: the state lives in VGPRs for the whole horizon, the nominal
// sequence (shift applied on read) sits in LDS, the standard normals stream in sample-minor
// rows-of-4 (one 1 KiB coalesced read per wave instruction) or come from Philox in registers.
// Replaces mppi.py:407-417 (see include/mppi_amd.h).  Nothing of shape (K,T,nu) is written.
//
// Memory pipeline (the part that decides the HBM fraction): at K = 65536 the chip holds ONE wave
// per SIMD, so latency must be covered by loads in flight from that wave alone.  The noise rows
// go through a register ring of D super-steps (9 rows-of-4 = 144 B per lane outstanding).
// gfx950 retires vector-memory ops in order behind ONE counter (vmcnt), and the compiler can
// only wait for "all but the N youngest": any load or store under a branch between two ring
// slots makes N unknowable and every wait collapses to vmcnt(0), which serialises the ring.
// Hence the main streaming loop contains NO branch and NO conditional vector-memory instruction:
// it runs whole groups of D complete super-steps, refills are unconditional (row index clamped),
// the horizon tail is consumed from the ring after the loop, and the rare work that needs extra
// memory traffic (sampler rows, `states` stores) lives in a separate instantiation (SLOW) chosen
// per wave.
#pragma once
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include "actions.hpp"
#include "dispatch.hpp"
#include "models.hpp"
#include "weights.hpp"

namespace mppi {

// Per-block tables in LDS, one entry per (t,n) of the horizon (J = T*nu each):
//   Ue[j]  base sequence: nominal U with the shift applied (mppi.py:232-238); SMPPI: A + U*dt
//   Um[j]  Ue[j] + mu[n]                    -> v = fma(z, sqrt(diag), Um)   (:201-206, :380)
//   G[j]   lambda * (Sigma^-1 U[t])[n]      -> pert += G * e                (:186-199, :415)
//          (always from the true nominal U, also when the base is something else)
// G folds lambda*e*Sigma^-1 . U into ONE fma per control dimension (for a full Sigma it removes
// the nu x nu product from the time loop entirely); the association differs from the
// reference's ((lambda*e)*Sigma^-1)*U by rounding only (parity tests: 1e-5 fp32 / 1e-9 fp64).
template <typename T>
struct StepTables {
  const T* Ue;
  const T* Um;
  const T* G;
  T* ktn_lds;      // MPPI_NOISE_KTN: wave-private transposition tiles (else unused)
  int kwave0;      // raw index of the first sample of this wave's current chunk
};

// one timestep: actions from z, action cost, dynamics, running cost
// SMOOTH = false: the caller knows this is not an SMPPI problem -- no smoothness cost (no uniform branch on
// a.smooth_w in the step) and e_scale == 1 (no rescaling multiply)
template <class Model, typename T, int NOISE, bool DIAG, int SLOW, bool SMOOTH = true>
__device__ __forceinline__ void rollout_step(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                             const Model& model, const StepTables<T>& tb, int k,
                                             bool active, int orow, int t, const T* zt,
                                             T (&x)[Model::NX], T (&vprev)[Model::NU], T& rollout,
                                             T& pert) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr bool SRC_ACTIONS = NOISE == MPPI_NOISE_ACTIONS;
  T z[NU], v[NU], u[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) z[n] = zt[n];
#ifdef MPPI_K1_STREAM_ONLY   // experiment: memory pipeline only (tools/k1_sweep.py), never shipped
#pragma unroll
  for (int n = 0; n < NU; ++n) rollout += z[n];
  return;
#endif
  const T* __restrict__ Ut = tb.Ue + t * NU;
  const T* __restrict__ Umt = tb.Um + t * NU;
  const T* __restrict__ Gt = tb.G + t * NU;
  if constexpr (SRC_ACTIONS) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = z[n];
  } else if constexpr (DIAG) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = m_fma(z[n], ac.sd[n], Umt[n]);
  } else {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      T Lr[NU];
      chol_row<T, NU>(ac.Lm, n, Lr);
      T s = Umt[n];
#pragma unroll
      for (int m = 0; m <= n; ++m) s = m_fma(z[m], Lr[m], s);               // L is lower triangular
      v[n] = s;
    }
  }
  if constexpr (SLOW == 1) {
    // null-action row only: a per-lane select, no memory traffic -> the ring's exact waits survive
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = orow == -1 ? T(0) : v[n];          // mppi.py:390-392
  } else if constexpr (SLOW == 2) {
    if (orow == -1) {
#pragma unroll
      for (int n = 0; n < NU; ++n) v[n] = T(0);                            // mppi.py:390-392
    } else if (orow >= 0) {
      const T* __restrict__ srow = a.sampler + ((long long)orow * a.Tn + t) * NU;
#pragma unroll
      for (int n = 0; n < NU; ++n) v[n] = srow[n];                         // :393-399
    }
  }
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    v[n] = clampT(v[n], ac.lo[n], ac.hi[n]);                               // :383
    const T e = SMOOTH ? (v[n] - Ut[n]) * ac.e_scale : v[n] - Ut[n];       // :385 (SMPPI :544)
    pert = m_fma(Gt[n], ac.abs_cost ? m_abs(e) : e, pert);                 // :409, :415
    u[n] = a.u_scale * v[n];                                               // :313
  }
  if (SMOOTH && a.smooth_w != T(0)) {
    // SMPPI smoothness cost w * |u_scale * (v[t] - v[t-1])|^2 (mppi.py:559-562); vprev = v at t = 0
    T d2 = T(0);
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      const T d = v[n] - vprev[n];
      d2 = m_fma(d, d, d2);
      vprev[n] = v[n];
    }
    if (t > 0) rollout = m_fma(a.smooth_w, d2, rollout);
  }
  model.step(x, u, t);                                                     // :314
  rollout += model.cost(x, u, t);                                          // :318-319
  if constexpr (SLOW == 2) {
    if (a.states != nullptr && active) {
      T* __restrict__ so = a.states + ((long long)k * a.Tn + t) * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) so[i] = x[i];                           // :321
    }
  }
}

// ---------------------------------------------------------------------------------------------
// M > 1 state rollouts per action sequence (mppi.py:334-373): the lane keeps M copies of the state,
// feeds every copy the same bounded action, lets the model's process noise tell them apart and
// accumulates   cost_samples[m] += c_m,   cost_var += var_m(c_m) * discount^t   (unbiased variance,
// torch's .var(dim=0));   total = mean_m(cost_samples[m] + terminal_m) + rollout_var_cost * cost_var.
// The action rows are read ONCE per sample whatever M is (the reference expands them M-fold), so the
// memory side is that of M = 1 and the extra work is arithmetic -- plain row reads, no register ring.
// Process noise: x[i] += sd[i] * n,  n ~ N(0,1) from the engine's Philox with its own key (seed ^ tag)
// and counter (sample, (t * MM + m) * ceil(NX/4) + block, command) -- oracle/philox.py restates it.
// ---------------------------------------------------------------------------------------------
// (PROCESS_NOISE_KEY_TAG / PROCESS_NOISE_MM: common.hpp -- mppi_process_noise_export restates the counter scheme)

template <class Model, typename T, int NOISE, bool DIAG, int MM>
__device__ __forceinline__ void rollout_stream_multi(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                                     const Model& model, const StepTables<T>& tb, int k, bool active,
                                                     int orow, const T (&x0)[Model::NX], T& rollout, T& pert) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT, NXB = (NX + 3) / 4;
  const int M = a.M;
  T xm[MM][NX], cs[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    cs[m] = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) xm[m][i] = x0[i];
  }
  T cvar = T(0), dpow = T(1);
  // SMPPI (mppi.py:551-554, :561): the smoothness cost belongs to the action sequence, not to a state rollout -- added once, behind
  // the mean over the M rollouts, exactly as the single-rollout kernels add it (rollout_step)
  T smooth = T(0), vprev[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) vprev[n] = T(0);
  const T inv_M = T(1) / (T)M, inv_Mm1 = T(1) / (T)(M - 1);
  const int nss = (a.Tn + TT - 1) / TT;
  for (int ss = 0; ss < nss; ++ss) {
    T zc[P4 * 4];
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      T r[4];
      noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
      zc[4 * i] = r[0]; zc[4 * i + 1] = r[1]; zc[4 * i + 2] = r[2]; zc[4 * i + 3] = r[3];
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = ss * TT + tt;
      if (t >= a.Tn) break;
      // ---- the action of this step: as rollout_step ----
      T z[NU], v[NU], e[NU], u[NU];
#pragma unroll
      for (int n = 0; n < NU; ++n) z[n] = zc[tt * NU + n];
      const T* __restrict__ srow = orow >= 0 ? a.sampler + ((long long)orow * a.Tn + t) * NU : nullptr;
      if (a.noise_src == MPPI_NOISE_ACTIONS) make_action<T, NU, true, true>(ac, tb.Ue + t * NU, srow, z, orow, v, e);
      else make_action<T, NU, DIAG, false>(ac, tb.Ue + t * NU, srow, z, orow, v, e);
      const T* __restrict__ Gt = tb.G + t * NU;
      if (a.smooth_w != T(0)) {
        T d2 = T(0);
#pragma unroll
        for (int n = 0; n < NU; ++n) {
          const T d = v[n] - vprev[n];
          d2 = m_fma(d, d, d2);
          vprev[n] = v[n];
        }
        if (t > 0) smooth = m_fma(a.smooth_w, d2, smooth);
      }
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        pert = m_fma(Gt[n], ac.abs_cost ? m_abs(e[n]) : e[n], pert);              // mppi.py:409, :415
        u[n] = a.u_scale * v[n];                                                  // :354
      }
      // ---- M copies of the state ----
      T c[MM], mean = T(0);
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        c[m] = T(0);
        if (m < M) {
          model.step(xm[m], u, t);                                                // :356
          if (a.proc_sd != nullptr) {
#pragma unroll
            for (int q = 0; q < NXB; ++q) {
              T w[4];
              static_assert(MM == PROCESS_NOISE_MM, "the process-noise counter scheme is defined for MM = PROCESS_NOISE_MM copies");
              philox_normal4<T>(a.seed ^ PROCESS_NOISE_KEY_TAG, a.call, a.k_offset + k, ((long long)t * MM + m) * NXB + q, w, a.seven != 0);
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (4 * q + i < NX) xm[m][4 * q + i] = m_fma(a.proc_sd[4 * q + i], w[i], xm[m][4 * q + i]);
            }
          }
          c[m] = model.cost(xm[m], u, t);                                         // :361
          cs[m] += c[m];                                                          // :362
          mean += c[m];
          if (a.states != nullptr && active) {                                    // :366, layout (M,K,T,nx)
            T* __restrict__ so = a.states + (((long long)m * a.K + k) * a.Tn + t) * NX;
#pragma unroll
            for (int i = 0; i < NX; ++i) so[i] = xm[m][i];
          }
        }
      }
      mean *= inv_M;
      T var = T(0);
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const T d = c[m] - mean;
        if (m < M) var = m_fma(d, d, var);
      }
      cvar = m_fma(var * inv_Mm1, dpow, cvar);                                    // :363-364
      dpow *= a.var_disc;
    }
  }
  T tot = T(0);
#pragma unroll
  for (int m = 0; m < MM; ++m)
    if (m < M) tot += cs[m] + (a.use_terminal ? model.terminal(xm[m]) : T(0));    // :369-370
  rollout = tot * inv_M + a.var_cost * cvar + smooth;                             // :371-372 (+ SMPPI :561)
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "csrc/rollout.hpp: the single-launch command's publish / ticket protocol and the inline CDNA4 assembly are written for gfx950 only"
#endif
#ifndef MPPI_K1_BLOCK
#define MPPI_K1_BLOCK 256   // threads per K1 workgroup (multiple of 64)
#endif
constexpr int K1_BLOCK = MPPI_K1_BLOCK;
// tools/micro/fuse_parts.hip: where the single-launch command's time goes (phase stamps of every workgroup on the device clock);
// compiled out of the product
#ifdef MPPI_FUSE_STAMPS
__device__ unsigned long long g_fuse_stamps[64 * 8];
#define MPPI_FUSE_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_fuse_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MPPI_FUSE_STAMP(i) do { } while (0)
#endif

#ifndef MPPI_K1_ROWS
// rows-of-4 (16 B each) a lane keeps in flight; fp64 uses half.  Measured at C3 (tools/run_k1_rows.sh,
// K1 inside the command pipeline): 24 rows 42.1 us | 16: 36.4 | 12: 37.6 | 9: 35.9 | 6: 39.2.  A deeper
// ring does NOT help: beyond ~150 B per lane in flight the kernel is not latency-bound any more,
// and the extra 60+ VGPRs push the allocator into AGPR copies inside the loop.
#define MPPI_K1_ROWS 9
#endif

// depth (in super-steps) of the register ring for a given control width / element type
template <int NU, typename T>
struct Ring {
  static constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  static constexpr int DROWS = (sizeof(T) == 4 ? MPPI_K1_ROWS : MPPI_K1_ROWS / 2) / P4;
  static constexpr int DSTEP = 16 / TT;                  // at most 16 timesteps ahead
  static constexpr int D0 = DROWS < DSTEP ? DROWS : DSTEP;
  static constexpr int D = D0 < 2 ? 2 : D0;
};

// MPPI_NOISE_KTN: the reference's (K,T,nu) row-major draw read in place.  Lane k needs its own
// contiguous stream, which one-lane-per-sample loads would fetch as 64 separate cache lines per
// instruction.  Instead a wave reads 128-B lines cooperatively (8 lanes x 16 B per sample line,
// 8 sample lines per instruction), transposes them through a wave-private LDS tile (XOR-swizzled
// 16-B slots: conflict-free on the write and on the read side) and ends up with exactly the
// rows-of-4 the TNK4 path would have read.  A macro-step = PL lines = MS whole timesteps.
template <int NU>
struct Ktn {
  static constexpr int G = (NU % 32 == 0) ? 32 : (NU % 16 == 0) ? 16 : (NU % 8 == 0) ? 8 : (NU % 4 == 0) ? 4 : (NU % 2 == 0) ? 2 : 1;
  static constexpr int PL = NU / G;            // 128-B lines (32 floats) per macro-step
  static constexpr int MS = 32 * PL / NU;      // timesteps per macro-step
  static constexpr bool OK = NU % 4 == 0 && MS <= 16 && PL <= 3;   // nu in {4, 8, 12, 16, 32}
  static constexpr int LDS_FLOATS_PER_WAVE = PL * 64 * 32;
};

// the PL x 8 coalesced 16-B loads of macro-step m: instruction (q,i) covers 8 sample lines
template <int NU, typename T>
__device__ __forceinline__ void ktn_issue(const KArgs<T>& a, int m, int kwave0, int c, int sl,
                                          float (&lreg)[Ktn<NU>::PL][8][4]) {
  constexpr int PL = Ktn<NU>::PL;
#pragma unroll
  for (int q = 0; q < PL; ++q) {
    long long off = (long long)(m * PL + q) * 32 + c * 4;        // float offset inside the sample's stream
    off = off + 4 <= a.J ? off : (long long)a.J - 4;              // tail: stay inside the row (masked by t < T)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int ks = kwave0 + 8 * i + sl;
      ks = ks < a.K ? ks : a.K - 1;
      load4<float>(reinterpret_cast<const float*>(a.z) + (long long)ks * a.J + off, 0, 0, 0, lreg[q][i]);
    }
  }
}

template <typename T, int NOISE, int NU>
__device__ __forceinline__ void ring_fetch(const KArgs<T>& a, int ss, int k, T (&dst)[Stream<NU>::P4 * 4]) {
  constexpr int P4 = Stream<NU>::P4;
#pragma unroll
  for (int i = 0; i < P4; ++i) {
    T r[4];
    noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
    dst[4 * i + 0] = r[0]; dst[4 * i + 1] = r[1]; dst[4 * i + 2] = r[2]; dst[4 * i + 3] = r[3];
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA ring (fp32, TNK4 / ACTIONS streams).  The register ring above keeps 9 rows (144 B per
// lane) in flight because every row in flight costs 4 VGPRs; HBM-cold that is not enough for ONE
// wave per SIMD to cover the memory latency (37.6 us = 5.36 TB/s at C3).  Here the rows travel
// global -> LDS directly (`global_load_lds_dwordx4`: one wave instruction moves the 1 KiB row
// [jb][k0..k0+63][4] -- contiguous in the sample-minor layout -- to LDS base + lane*16), so depth
// costs LDS instead of registers: DMA_ROWS rows per wave (30 KiB) in flight, 120 KiB per CU.
// hipcc does not count asm memory operations, so the waits are ours: the main loop keeps exactly
// D slots (of P4 rows) outstanding and consumes the oldest behind `s_waitcnt vmcnt((D-1)*P4)`;
// the compiler's own loads/stores only make those waits more conservative (vector memory retires
// in order).  A slot is refilled right after it was read (`lgkmcnt(0)` first: the ds_reads have
// returned before the DMA that overwrites them is issued).
// ---------------------------------------------------------------------------------------------
template <int NU, int ROWS_MAX>
struct DmaRing {
  static constexpr int P4 = Stream<NU>::P4;
  static constexpr int D = (ROWS_MAX / P4) < 2 ? 2 : (ROWS_MAX / P4);   // slots (super-steps) in flight
  static constexpr int ROWS = D * P4;
  static constexpr int FLOATS_PER_WAVE = ROWS * 256;
  static constexpr bool OK = ROWS <= 63;                                 // vmcnt is a 6-bit counter
};

// one row-of-4 for the 64 samples of this wave: global (uniform row base + per-lane byte offset) ->
// LDS (uniform byte address + lane*16).  M0 carries the LDS address and is compiler-reserved: saved
// and restored inside the statement.
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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class Model, typename T, int NOISE, bool DIAG, int SLOW, int ROWS_MAX>
__device__ __forceinline__ void rollout_stream_dma(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                                   const Model& model, const StepTables<T>& tb, int k,
                                                   bool active, int orow, float* ring_wave,
                                                   T (&x)[Model::NX], T& rollout, T& pert) {
  static_assert(sizeof(T) == 4 && (NOISE == MPPI_NOISE_TNK4 || NOISE == MPPI_NOISE_ACTIONS), "fp32 row streams only");
  constexpr int NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  constexpr int D = DmaRing<NU, ROWS_MAX>::D;
  const int nss = (a.Tn + TT - 1) / TT;
  const int last = nss - 1;
  const int lane = threadIdx.x & (WAVE - 1);
  const unsigned voff = (unsigned)k * 16u;                     // this lane's sample inside a row (bytes)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)ring_wave);
  const float* zbase = reinterpret_cast<const float*>(a.z);
  const long long row_floats = a.zp * 4;
  const float* my = ring_wave + lane * 4;
  T vprev[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) vprev[n] = T(0);

  auto issue = [&](int ss_src, int slot) {
#pragma unroll
    for (int i = 0; i < P4; ++i)
      dma_row16(zbase + ((long long)ss_src * P4 + i) * row_floats, voff, lds0 + (unsigned)((slot * P4 + i) * 1024));
  };
  // Every load the compiler knows about (per-control constants, initial state) is waited for HERE,
  // with a wait the compiler sees: otherwise it defers those waits to the first use inside the
  // loop, as vmcnt(0) -- which also drains the freshly issued ring (the asm loads are invisible to
  // its bookkeeping, but vector memory retires in order).
  __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), expcnt/lgkmcnt untouched
  // prologue: D slots in flight (short horizons re-read the last super-step; never consumed)
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d < last ? d : last, d);

  int slot = 0;
  const int nmain = nss > D ? nss - D : 0;
  for (int ss = 0; ss < nmain; ++ss) {
    wait_vmcnt<(D - 1) * P4>();                               // the oldest slot has landed
    T zt[P4 * 4];
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      const float4 v4 = *reinterpret_cast<const float4*>(my + (slot * P4 + i) * 256);
      zt[4 * i] = v4.x; zt[4 * i + 1] = v4.y; zt[4 * i + 2] = v4.z; zt[4 * i + 3] = v4.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // slot read out: safe to overwrite
    issue(ss + D, slot);                                      // ss + D <= last here
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, tb, k, active, orow, ss * TT + tt, &zt[tt * NU], x, vprev,
                                                rollout, pert);
    slot = slot + 1 == D ? 0 : slot + 1;
  }
  // tail: the ring holds super-steps nmain .. nmain+D-1 in issue order, nothing is refilled
  static_for<0, D>([&](auto dc) {
    constexpr int d = decltype(dc)::value;
    const int ss = nmain + d;
    if (ss < nss) {                                           // wave-uniform
      wait_vmcnt<(D - 1 - d) * P4>();
      T zt[P4 * 4];
#pragma unroll
      for (int i = 0; i < P4; ++i) {
        const float4 v4 = *reinterpret_cast<const float4*>(my + (slot * P4 + i) * 256);
        zt[4 * i] = v4.x; zt[4 * i + 1] = v4.y; zt[4 * i + 2] = v4.z; zt[4 * i + 3] = v4.w;
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int t = ss * TT + tt;
        if (t < a.Tn)
          rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, tb, k, active, orow, t, &zt[tt * NU], x, vprev,
                                                    rollout, pert);
      }
      slot = slot + 1 == D ? 0 : slot + 1;
    }
  });
  wait_vmcnt<0>();   // no DMA may still be in flight when the wave (and its LDS allocation) ends
}

template <class Model, typename T, int NOISE, bool DIAG, int SLOW>
__device__ __forceinline__ void rollout_stream(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                               const Model& model, const StepTables<T>& tb, int k,
                                               bool active, int orow,
                                               T (&ring)[Ring<Model::NU, T>::D][Stream<Model::NU>::P4 * 4],
                                               T (&x)[Model::NX], T& rollout, T& pert) {
  T vprev[Model::NU];
#pragma unroll
  for (int n = 0; n < Model::NU; ++n) vprev[n] = T(0);
  constexpr int NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  const int nss = (a.Tn + TT - 1) / TT;

  if constexpr (NOISE == MPPI_NOISE_KTN) {
    if constexpr (Ktn<NU>::OK && sizeof(T) == 4) {
      constexpr int PL = Ktn<NU>::PL, MS = Ktn<NU>::MS;
      const int lane = threadIdx.x & (WAVE - 1);
      const int kwave0 = tb.kwave0;                     // first sample of this wave (raw, not clamped)
      float* ldsw = reinterpret_cast<float*>(tb.ktn_lds) + (threadIdx.x / WAVE) * Ktn<NU>::LDS_FLOATS_PER_WAVE;
      const int nmacro = (a.Tn + MS - 1) / MS;
      const int c = lane & 7, sl = lane >> 3;
      float lreg[PL][8][4];      // plain scalars: a float4 array is not promoted to registers here
      ktn_issue<NU>(a, 0, kwave0, c, sl, lreg);
      for (int m = 0; m < nmacro; ++m) {
        // stage macro-step m: registers -> swizzled LDS tile (wave-private: LDS ops of one wave execute
        // in order, no barrier).  The tile then serves the MS steps directly (NU/4 ds_read_b128 each),
        // so the only long-lived registers are the lines of macro-step m+1 in flight.
#pragma unroll
        for (int q = 0; q < PL; ++q) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int s = 8 * i + sl;
            store4<float>(ldsw + q * 2048, 0, 0, 8 * s + (c ^ (s & 7)), lreg[q][i]);
          }
        }
        ktn_issue<NU>(a, m + 1 < nmacro ? m + 1 : m, kwave0, c, sl, lreg);   // unconditional prefetch of the next macro-step
#pragma unroll
        for (int st = 0; st < MS; ++st) {
          const int t = m * MS + st;
          if (t < a.Tn) {
            T zt[NU];
#pragma unroll
            for (int r = 0; r < NU / 4; ++r) {
              const int f = st * NU + 4 * r;               // float offset inside the macro-step (static)
              const int q = f / 32, c2 = (f % 32) / 4;
              const float4 v4 = *reinterpret_cast<const float4*>(ldsw + q * 2048 + (8 * lane + (c2 ^ (lane & 7))) * 4);
              zt[4 * r] = v4.x; zt[4 * r + 1] = v4.y; zt[4 * r + 2] = v4.z; zt[4 * r + 3] = v4.w;
            }
            rollout_step<Model, T, MPPI_NOISE_TNK4, DIAG, SLOW>(a, ac, model, tb, k, active, orow, t, zt, x, vprev,
                                                                rollout, pert);
          }
        }
      }
    }
    return;
  } else if constexpr (NOISE == MPPI_NOISE_PHILOX) {
    // Generated in registers: no memory pipeline, but Philox4x32-10 is a chain of 10 dependent
    // 32x32->64 multiplies and this wave is alone on its SIMD, so PB super-steps are generated
    // together (independent counters -> the chains interleave) before they are consumed.
    constexpr int PB0 = 16 / (P4 * 4) > 0 ? 64 / (P4 * 4) : 1;       // ~64 normals per batch
    constexpr int PB = PB0 < 2 ? 2 : (PB0 > 8 ? 8 : PB0);
    for (int ss0 = 0; ss0 < nss; ss0 += PB) {
      T zb[PB][P4 * 4];
#pragma unroll
      for (int b = 0; b < PB; ++b) {
#pragma unroll
        for (int i = 0; i < P4; ++i) {
          T r[4];
          const long long jb = (long long)(ss0 + b) * P4 + i;
          noise4<T, NOISE>(a, jb, k, r);   // rows past the horizon: unused
          // "generate once": keep the stream for K3 to re-read instead of regenerating it
          if (a.z != nullptr && active && jb < a.J4) store4<T>(const_cast<T*>(a.z), a.zp, jb, k, r);
          zb[b][4 * i + 0] = r[0]; zb[b][4 * i + 1] = r[1]; zb[b][4 * i + 2] = r[2]; zb[b][4 * i + 3] = r[3];
        }
      }
#pragma unroll
      for (int b = 0; b < PB; ++b) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int t = (ss0 + b) * TT + tt;
          if (t < a.Tn)
            rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, tb, k, active, orow, t, &zb[b][tt * NU], x,
                                                      vprev, rollout, pert);
        }
      }
    }
    return;
  } else {
    constexpr int D = Ring<NU, T>::D;
    const int last = nss - 1;
    const int nss_full = a.Tn / TT;          // super-steps whose TT timesteps all exist
    int kf = k;                              // sample index as seen by the refill loads
    auto fetch = [&](int ss, T (&dst)[P4 * 4]) { ring_fetch<T, NOISE, NU>(a, ss, kf, dst); };

    int ss0 = 0;
    // ---- main loop: whole groups of D complete super-steps, no branch, no conditional VMEM ----
    for (; ss0 + D <= nss_full; ss0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int ss = ss0 + d;
        // First use of slot d is pinned behind the previous step's result: the scheduler would
        // otherwise start the (state-independent) colouring arithmetic of all D slots early,
        // and the wait for the youngest of them drains the ring.
#pragma unroll
        for (int i = 0; i < P4 * 4; ++i) asm volatile("" : "+v"(ring[d][i]) : "v"(rollout));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          int t = ss * TT + tt;
          // same pin for the LDS table reads of this step (their addresses depend only on t)
          asm volatile("" : "+s"(t) : "v"(rollout));
          rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, tb, k, active, orow, t,
                                                    &ring[d][tt * NU], x, vprev, rollout, pert);
        }
        // Refill slot d only AFTER it has been consumed (the address is made to depend on the
        // step's result): the load then lands in the same registers, the loop carries no copy of
        // a just-loaded value and therefore no wait on the youngest loads at the back edge.
        asm volatile("" : "+v"(kf) : "v"(rollout));
        fetch(ss + D < last ? ss + D : last, ring[d]);
      }
    }
    // ---- tail: the ring already holds rows ss0 .. ss0+D-1; consume what exists ----
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int ss = ss0 + d;
      if (ss < nss) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int t = ss * TT + tt;
          if (t < a.Tn)
            rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, tb, k, active, orow, t,
                                                      &ring[d][tt * NU], x, vprev, rollout, pert);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// HEAVY models (`static constexpr bool HEAVY = true`: traced networks, jit.from_torch -- a step of thousands of
// instructions).  The streams above unroll the step once per ring slot and timestep of a super-step, in three per-wave
// variants: dozens of copies per kernel, which is right for a 20-instruction integrator and wrong here (minutes of
// hipcc, and a loop body far beyond the instruction cache).  This stream keeps ONE copy of the step: a rolled loop over
// super-steps and timesteps, the timestep's controls picked out of the row registers by selects, the next rows one
// super-step ahead (their latency hides behind TT heavy steps), the general per-lane variant (sampler rows, states).
// ---------------------------------------------------------------------------------------------
template <class M, class = void>
struct model_heavy : std::false_type {};
template <class M>
struct model_heavy<M, std::enable_if_t<M::HEAVY>> : std::true_type {};

template <class Model, typename T, int NOISE, bool DIAG>
__device__ __forceinline__ void rollout_stream_heavy(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                                     const Model& model, const StepTables<T>& tb, int k, bool active,
                                                     int orow, T (&x)[Model::NX], T& rollout, T& pert) {
  static_assert(NOISE != MPPI_NOISE_KTN, "heavy models take their rows in the engine's layout");
  constexpr int NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  T vprev[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) vprev[n] = T(0);
  const int nss = (a.Tn + TT - 1) / TT, last = nss - 1;
  T cur[P4 * 4], nxt[P4 * 4];
  auto get = [&](int ss, T (&dst)[P4 * 4]) {
    if constexpr (NOISE == MPPI_NOISE_PHILOX) {
#pragma unroll
      for (int i = 0; i < P4; ++i) {
        T r[4];
        const long long jb = (long long)ss * P4 + i;
        noise4<T, NOISE>(a, jb, k, r);
        if (a.z != nullptr && active && jb < a.J4) store4<T>(const_cast<T*>(a.z), a.zp, jb, k, r);   // "generate once"
        dst[4 * i + 0] = r[0]; dst[4 * i + 1] = r[1]; dst[4 * i + 2] = r[2]; dst[4 * i + 3] = r[3];
      }
    } else {
      ring_fetch<T, NOISE, NU>(a, ss, k, dst);
    }
  };
  get(0, cur);
#pragma nounroll
  for (int ss = 0; ss < nss; ++ss) {
    if constexpr (NOISE != MPPI_NOISE_PHILOX) get(ss < last ? ss + 1 : last, nxt);
#pragma nounroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = ss * TT + tt;
      if (t >= a.Tn) break;                                    // wave-uniform
      T zt[NU];
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        T v = cur[n];
        static_for<1, TT>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          v = tt == j ? cur[j * NU + n] : v;
        });
        zt[n] = v;
      }
      rollout_step<Model, T, NOISE, DIAG, 2>(a, ac, model, tb, k, active, orow, t, zt, x, vprev, rollout, pert);
    }
    if constexpr (NOISE != MPPI_NOISE_PHILOX) {
#pragma unroll
      for (int i = 0; i < P4 * 4; ++i) cur[i] = nxt[i];
    } else {
      if (ss < last) get(ss + 1, cur);
    }
  }
}

// NOISE: MPPI_NOISE_TNK4 | _PHILOX | _ACTIONS (compile-time);  DIAG: diagonal Sigma;
// DMA_ROWS > 0: the rows travel through the LDS-DMA ring (fp32 row streams), 0: register ring
// FUSE: the whole command of a small problem in this ONE launch (see the block behind the chunk loop)
// MM > 1: up to MM state rollouts per action sequence (rollout_stream_multi)
template <class Model, typename T, int NOISE, bool DIAG, int DMA_ROWS = 0, bool FUSE = false, int MM = 1>
__global__ void __launch_bounds__(K1_BLOCK) rollout_cost_kernel(const KArgs<T> a_in) {
  constexpr int NX = Model::NX, NU = Model::NU;
  const KArgs<T> a = env_view(a_in);        // MPPI_Batched: environment = blockIdx.z
  stamp_entry(a.tstamp);
  MPPI_FUSE_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);   // [J] nominal sequence, shift applied
  T* Um = Ue + a.J;                         // [J] Ue + mu
  T* G = Um + a.J;                          // [J] lambda * Sigma^-1 Ue[t]
  T* red = G + a.J;                         // [BLOCK/WAVE]
  T* fac = red + BLOCK / WAVE;              // [2*NU*NU] full-Sigma factors (only if !DIAG)

  // A workgroup handles the 256-sample chunks blockIdx.x, blockIdx.x + gridDim.x, ... (persistent
  // form for K beyond one chunk per CU: every resident workgroup walks the rows in step with the
  // others, chunk after chunk, instead of four desynchronised waves of workgroups)
  int kraw = blockIdx.x * K1_BLOCK + threadIdx.x;
  bool active = kraw < a.K;
  int k = active ? kraw : a.K - 1;         // tail lanes shadow the last sample, never store
  int orow = overwrite_row(a, a.k_offset + k);

  // Issue order matters (loads retire in order): first the few loads the set-up needs (nominal
  // sequence, initial state), then the ring prologue, so that the set-up's waits do not sit
  // behind 24 KiB of noise per wave and the noise latency overlaps the LDS fill + barrier.
  constexpr int UL = 4;     // staged in registers: J <= 4*K1_BLOCK (e.g. T=64, nu=12 -> 3 per thread)
  T uload[UL], umload[UL], gload[UL];
#pragma unroll
  for (int q = 0; q < UL; ++q) {
    const int j = threadIdx.x + q * K1_BLOCK;
    const bool ok = j < a.J;
    const int n = ok ? j % NU : 0;
    uload[q] = ok ? u_base(a, j) : T(0);
    umload[q] = (ok && !a.coloured) ? a.mu[n] : T(0);
    // diagonal Sigma: G = lambda * U / sigma^2 needs no neighbours -> built in the same pass
    if constexpr (DIAG) gload[q] = ok ? (a.B != nullptr ? u_eff(a, j) : uload[q]) * a.sinv[n * NU + n] : T(0);
  }
  const Model model(a);
  T x[NX];
  {
    const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = s0[i];      // mppi.py:302-305
  }
  ActionConsts<T, NU> ac;
  ac.load(a, DIAG ? nullptr : fac);

  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT, D = Ring<NU, T>::D;
  T ring[D][P4 * 4];
  constexpr bool HEAVY = model_heavy<Model>::value;
  auto ring_prologue = [&]() {
    if constexpr (NOISE != MPPI_NOISE_PHILOX && NOISE != MPPI_NOISE_KTN && !HEAVY) {
      const int last = (a.Tn + TT - 1) / TT - 1;
#pragma unroll
      for (int d = 0; d < D; ++d) ring_fetch<T, NOISE, NU>(a, d < last ? d : last, k, ring[d]);
    }
  };
  if constexpr (DMA_ROWS == 0 && MM == 1) ring_prologue();

#pragma unroll
  for (int q = 0; q < UL; ++q) {
    const int j = threadIdx.x + q * K1_BLOCK;
    if (j < a.J) {
      Ue[j] = uload[q];
      Um[j] = uload[q] + umload[q];
      if constexpr (DIAG) G[j] = a.lambda_ * gload[q];
    }
  }
  for (int j = threadIdx.x + UL * K1_BLOCK; j < a.J; j += K1_BLOCK) {   // very long horizons
    const int n = j % NU;
    const T ub = u_base(a, j);
    Ue[j] = ub;
    Um[j] = a.coloured ? ub : ub + a.mu[n];
    if constexpr (DIAG) G[j] = a.lambda_ * (u_eff(a, j) * a.sinv[n * NU + n]);
  }
  const bool coloured_full = DIAG && a.coloured && !a.diag;
  if (coloured_full) {
    // generator-coloured stream of a full Sigma: this (diagonal) form streams it, but the action
    // cost is still lambda * eps' Sigma^-1 U.  Sigma^-1 goes through LDS (the launch reserved NU*NU
    // elements at `fac`): one coalesced load per thread instead of every lane of every wave
    // hammering the same 576 bytes of global memory
    for (int i = threadIdx.x; i < NU * NU; i += K1_BLOCK) fac[i] = a.sinv[i];
  }
  __syncthreads();
  if (coloured_full) {
    for (int j = threadIdx.x; j < a.J; j += K1_BLOCK) {
      const int n = j % NU, t0 = j - n;
      T g = T(0);
#pragma unroll
      for (int m = 0; m < NU; ++m) g = m_fma(fac[n * NU + m], a.B != nullptr ? u_eff(a, t0 + m) : Ue[t0 + m], g);
      G[j] = a.lambda_ * g;
    }
    __syncthreads();
  }
  if constexpr (!DIAG) {
    // full Sigma: G[t,n] = lambda * sum_m Sigma^-1[n,m] U[t,m] needs the whole row -> second pass
    for (int j = threadIdx.x; j < a.J; j += K1_BLOCK) {
      const int n = j % NU, t0 = j - n;
      T g = T(0);
      for (int m = 0; m < NU; ++m) g = m_fma(ac.Sm[n * NU + m], u_eff(a, t0 + m), g);   // Sigma^-1 symmetric
      G[j] = a.lambda_ * g;
    }
    __syncthreads();
  }
  // KTN tiles / DMA ring start 16-B aligned behind the tables and the factor block (present unless Sigma is diagonal)
  T* ktn_lds = fac + ((DIAG && a.diag) ? 0 : 2 * NU * NU);
  ktn_lds += (4 - ((ktn_lds - Ue) & 3)) & 3;
  StepTables<T> tb{Ue, Um, G, ktn_lds, kraw - (int)(threadIdx.x & (WAVE - 1))};

  const int nchunks = (a.K + K1_BLOCK - 1) / K1_BLOCK;
  T fuse_total = T(0);
  MPPI_FUSE_STAMP(1);
  for (int chunk = blockIdx.x;;) {
  T rollout = T(0), pert = T(0);
  // wave-uniform choice: does this wave own overwritten rows, or must it store the states?
  // (0 = neither; 1 = only the sample_null_action row: a select, no extra memory traffic; 2 = sampler
  // rows / states: conditional loads and stores, which cost the wave its exact vmcnt waits)
  const bool slow = __any(orow >= 0) || a.states != nullptr;
  if constexpr (MM > 1) {
    rollout_stream_multi<Model, T, NOISE, DIAG, MM>(a, ac, model, tb, k, active, orow, x, rollout, pert);
  } else if constexpr (DMA_ROWS > 0) {
    // wave-private ring behind the tables (1 KiB rows; the carve keeps it 16-B aligned)
    float* ring_wave = reinterpret_cast<float*>(ktn_lds) + (threadIdx.x / WAVE) * DmaRing<NU, DMA_ROWS>::FLOATS_PER_WAVE;
    // (launch_rollout sends problems with sampler rows or a `states` output to the register-ring
    // kernel: their conditional vector-memory traffic is waited for with vmcnt(0) by the compiler,
    // which would drain a DMA ring every step)
    if (__any(orow == -1)) {
      rollout_stream_dma<Model, T, NOISE, DIAG, 1, DMA_ROWS>(a, ac, model, tb, k, active, orow, ring_wave, x, rollout, pert);
    } else {
      rollout_stream_dma<Model, T, NOISE, DIAG, 0, DMA_ROWS>(a, ac, model, tb, k, active, orow, ring_wave, x, rollout, pert);
    }
  } else if constexpr (HEAVY) {
    if constexpr (NOISE != MPPI_NOISE_KTN) rollout_stream_heavy<Model, T, NOISE, DIAG>(a, ac, model, tb, k, active, orow, x, rollout, pert);
  } else {
    if (slow)
      rollout_stream<Model, T, NOISE, DIAG, 2>(a, ac, model, tb, k, active, orow, ring, x, rollout, pert);
    else if (__any(orow == -1))
      rollout_stream<Model, T, NOISE, DIAG, 1>(a, ac, model, tb, k, active, orow, ring, x, rollout, pert);
    else
      rollout_stream<Model, T, NOISE, DIAG, 0>(a, ac, model, tb, k, active, orow, ring, x, rollout, pert);
  }

  if constexpr (MM == 1) {
    if (a.use_terminal) rollout += model.terminal(x);                  // :324-328
  }
  const T total = rollout + pert;                                      // :416
  if (active) {
    a.cost[k] = total;
    if (a.pert != nullptr) a.pert[k] = pert;
  }
  const T bm = wave_min<T>(active ? total : inf_v<T>());      // one minimum per 64 samples
  if ((threadIdx.x & (WAVE - 1)) == 0 && kraw < a.K) a.block_min[kraw / WAVE] = bm;
  if constexpr (FUSE) { fuse_total = total; break; }          // fused launches have one chunk per workgroup

  // ---- next chunk of this workgroup (the LDS tables stay) ----
  chunk += gridDim.x;
  if (chunk >= nchunks) break;
  kraw = chunk * K1_BLOCK + threadIdx.x;
  active = kraw < a.K;
  k = active ? kraw : a.K - 1;
  orow = overwrite_row(a, a.k_offset + k);
  tb.kwave0 = kraw - (int)(threadIdx.x & (WAVE - 1));
  {
    const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = s0[i];
  }
  if constexpr (DMA_ROWS == 0 && MM == 1) ring_prologue();
  }

  if constexpr (FUSE) {
    // ------------------------------------------------------------------------------------------
    // Single-launch command for small problems (K <= 64 workgroups, T*nu <= 256): a command of
    // 8192 x 32 is 1 MB of traffic and was three launches bound by their boundaries and by the host
    // (12.6 + 5 + 5 us of kernels inside a 31 us command).  Here the workgroup goes on, behind its
    // rollout, with
    //   B  its own part of K3 -- weights relative to the WORKGROUP's minimum beta_b, eta_b, and
    //      P_b[j] = sum_k w_k eps'_k[j] over its 256 samples (the rows it has just read or generated are
    //      re-read out of L1/L2) -- published as a partial record {beta_b, eta_b, P_b};
    //   C  (the LAST workgroup to finish, elected by an arrival ticket) the rank-order style combine
    //      of all partial records in workgroup order -- beta = min beta_b, s_b = exp(-(beta_b-beta)/lambda),
    //      eta = sum s_b eta_b, P = sum s_b P_b -- and K4's update U_out = shift(U) + P/eta, the action
    //      and the shard record.
    // No grid barrier and no spinning: nobody waits for anybody.  Partial records cross CUs / XCDs
    // through agent-scope (sc1, write-through / L1-bypassing) stores and loads on BOTH sides -- per-XCD
    // L2s are not coherent, and plain stores would need an agent release fence that also writes back
    // the freshly generated rows (~6 us).  The protocol therefore rests on gfx950's sc1 store / load
    // semantics and on the explicit vmcnt(0) in front of the ticket, not on the language memory model
    // (formally the relaxed accesses race): it is pinned to this architecture below and soaked by
    // tests/test_gpu_edge_semantics.py (10^4 commands, two controllers on two streams).  omega and
    // cost_total_non_zero are NOT written (the caller opted in by passing NULL for both): they are
    // exp(-(cost_total - record[0]) / lambda) [/ record[1]].  The ticket lives in the workspace and
    // is left at 0 again.
    // ------------------------------------------------------------------------------------------
    __shared__ __attribute__((aligned(16))) T f_cU[UPD_TJ], f_cS[UPD_TJ], f_cM[UPD_TJ], f_cLo[UPD_TJ], f_cHi[UPD_TJ];
    __shared__ T f_red[K1_BLOCK / WAVE];
    __shared__ T f_wsum[K1_BLOCK / WAVE][UPD_TJ];
    __shared__ T f_s[64];
    __shared__ int f_last;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const int b = blockIdx.x, nb = gridDim.x;
    const T inv_lambda = T(1) / a.lambda_;
    MPPI_FUSE_STAMP(2);
    const T beta_b = block_min<T>(active ? fuse_total : inf_v<T>(), f_red);
    const T wk = active ? weight_of<T>(fuse_total, beta_b, inv_lambda) : T(0);
    const T eta_b = block_sum<T>(wk, f_red);
    const bool over = a.null_action && (a.k_offset + kraw) == 0;          // the sample_null_action row (no sampler rows here)
    const T w1[1] = {over ? T(0) : wk};
    const bool live1[1] = {__ballot(w1[0] != T(0)) != 0ull};
    const int njt = a.Jpad / UPD_TJ;
    MPPI_FUSE_STAMP(3);
    for (int jt = 0; jt < njt; ++jt) {
      const int j0 = jt * UPD_TJ;
      __syncthreads();                       // the previous tile's constants and sums are consumed
      if (threadIdx.x < UPD_TJ) {
        const int j = j0 + threadIdx.x;
        const bool ok = j < a.J;
        const int n = ok ? j % NU : 0;
        f_cU[threadIdx.x] = ok ? u_base(a, j) : T(0);
        f_cS[threadIdx.x] = ok ? (a.coloured ? T(1) : a.L[n * NU + n]) : T(0);
        f_cM[threadIdx.x] = (ok && !a.coloured) ? a.mu[n] : T(0);
        f_cLo[threadIdx.x] = ok ? a.umin[n] : T(0);
        f_cHi[threadIdx.x] = ok ? a.umax[n] : T(0);
      }
      __syncthreads();
      T acc[UPD_TJ];
#pragma unroll
      for (int i = 0; i < UPD_TJ; ++i) acc[i] = T(0);
      const int nrows = a.J4 - jt * (UPD_TJ / 4);
      if (live1[0]) {
        // One wave per SIMD here (a small problem fills a fraction of the chip): the 16 row reads of the
        // tile are all issued before the first is used -- row by row, each would expose an L2 round trip
        T zz[UPD_TJ / 4][4];
#pragma unroll
        for (int jbl = 0; jbl < UPD_TJ / 4; ++jbl) {
          const int jr = jbl < nrows ? jbl : nrows - 1;                  // clamped: no branch between the loads
          load4<T>(a.z, a.zp, (long long)jt * (UPD_TJ / 4) + jr, k, zz[jbl]);
        }
#pragma unroll
        for (int jbl = 0; jbl < UPD_TJ / 4; ++jbl) {
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const int cc = 4 * jbl + c4;
            T v = f_cU[cc] + (zz[jbl][c4] * f_cS[cc] + f_cM[cc]);
            v = clampT(v, f_cLo[cc], f_cHi[cc]);
            acc[cc] = jbl < nrows ? w1[0] * (v - f_cU[cc]) : T(0);
          }
        }
      }
      f_wsum[wv][lane] = live1[0] ? wave_reduce_transpose64<T>(acc) : T(0);
      __syncthreads();
      if (threadIdx.x < UPD_TJ) {
        T sum = f_wsum[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < K1_BLOCK / WAVE; ++i) sum += f_wsum[i][threadIdx.x];
        const int j = j0 + threadIdx.x;
        if (a.null_action && a.k_offset == 0 && b == 0 && j < a.J) {
          // the overwritten row 0 (masked above): its noise is clamp(0) - U, weight from its own cost
          const T w0 = __shfl(wk, 0, WAVE);                              // wave 0 == threads < 64 here
          sum += w0 * (clampT(T(0), f_cLo[threadIdx.x], f_cHi[threadIdx.x]) - f_cU[threadIdx.x]);
        }
        if (j < a.Jpad)
          __hip_atomic_store(&a.P_part[(long long)b * a.Jpad + j], sum * a.e_scale, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store(&a.eta_part[b], eta_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.block_min[(long long)b * (K1_BLOCK / WAVE)], beta_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every partial store of this workgroup has left the CU before its ticket is drawn
    MPPI_FUSE_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f_last = old == (unsigned)(nb - 1);
    }
    __syncthreads();
    MPPI_FUSE_STAMP(5);
    if (f_last) {
      // ---- C: combine, in workgroup order whoever is last ----
      // All loads of the first pass are issued before anything waits for one of them: an agent-scope
      // load is a round trip to the memory side (~1-2 us), and beta -> s_b -> eta -> P as dependent
      // round trips was most of this phase's time.
      constexpr int SL = K1_BLOCK / UPD_TJ, MAXI = 64 / SL;              // 4 slices of the workgroup index, <= 16 each
      const int c = threadIdx.x & (UPD_TJ - 1), sl = threadIdx.x / UPD_TJ;
      T bmin = inf_v<T>(), etab = T(0);
      if (threadIdx.x < nb) {
        bmin = __hip_atomic_load(&a.block_min[(long long)threadIdx.x * (K1_BLOCK / WAVE)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        etab = __hip_atomic_load(&a.eta_part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      T pp[MAXI];
      auto load_tile = [&](int j) {
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
          const int i = sl + SL * q;
          pp[q] = (i < nb && j < a.J)
                      ? __hip_atomic_load(&a.P_part[(long long)i * a.Jpad + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : T(0);
        }
      };
      load_tile(c);
      const T my_beta = bmin;
      const T beta = block_min<T>(bmin, f_red);
      if (threadIdx.x < nb) {
        const T sb = m_exp(-inv_lambda * (my_beta - beta));
        f_s[threadIdx.x] = sb;
        f_cU[threadIdx.x] = sb * etab;
      }
      __syncthreads();
      T eta = T(0);
      for (int i = 0; i < nb; ++i) eta += f_cU[i];                        // fixed order
      const T inv_eta = T(1) / eta;
      // P[j] = sum_b s_b P_b[j]: 64 columns x 4 interleaved slices of the workgroup index per pass,
      // slices combined in fixed order
      for (int j0 = 0; j0 < a.J; j0 += UPD_TJ) {
        const int j = j0 + c;
        if (j0 > 0) load_tile(j);
        T part = T(0);
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
          const int i = sl + SL * q;
          part += (i < nb ? f_s[i] : T(0)) * pp[q];
        }
        f_wsum[sl][c] = part;
        __syncthreads();
        if (threadIdx.x < UPD_TJ && j < a.J) {
          T P = f_wsum[0][c];
#pragma unroll
          for (int q = 1; q < SL; ++q) P += f_wsum[q][c];
          a.record[2 + j] = P;
          if (a.fuse) {                                                  // apply: mppi.py:268-275
            const T un = u_eff(a, j) + P * inv_eta;
            a.U_out[j] = un;
            if (a.action_out != nullptr && j < a.u_per_command * a.nu) a.action_out[j] = un;
          }
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        a.record[0] = beta;
        a.record[1] = eta;
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next command
      }
      MPPI_FUSE_STAMP(6);
    }
  }
  if (a.tstamp != nullptr) {
    // exit stamp AFTER every wave of the workgroup is done (the stamp is the kernel's span on the
    // device clock; without the barrier it missed the three other waves of the last workgroup)
    __syncthreads();
    stamp_exit(a.tstamp);
  }
}

}  // namespace mppi
#include "rollout_kmppi.hpp"   // KMPPI: interpolation inside K1 (needs rollout_step)
#include "mlp_wide.hpp"        // traced models with dense layers: matrix-core execution, sixteen samples per wave (needs rollout_stream_heavy)
#include "rollout_onchip.hpp"  // rng="philox" without a (K,T,nu) array: generate, roll out, keep eps' on chip, partial records
#include "rollout_onchip_pair.hpp"  // the same command with two waves per sample group (a second wave per SIMD hides the first one's waits)
#include "rollout_copies.hpp" // M > 1 rollouts per action sequence: one wave per copy (needs Stream, make_action)
namespace mppi {

// LDS-DMA ring depth (rows of 1 KiB per wave) for this launch, 0 = register ring.  One workgroup
// per CU (<= 256 workgroups, the C3 case: one wave per SIMD) gets the deep ring (30 rows, 120 KiB
// per workgroup); larger grids keep two workgroups per CU resident with 15 rows each, so that one
// workgroup's prologue / epilogue overlaps the other's streaming.  MPPI_K1_DMA=0|15|30 overrides
// (tools/ A/B runs), read once.
template <class Model, typename T>
static int dma_rows_for(const KArgs<T>& a, size_t smem_tables) {
  if constexpr (sizeof(T) != 4) return 0;
  constexpr int NU = Model::NU;
  if (a.noise_src != MPPI_NOISE_TNK4 && a.noise_src != MPPI_NOISE_ACTIONS) return 0;
  if (a.states != nullptr || a.n_sampler > 0) return 0;    // conditional VMEM per step: register-ring kernel
#ifndef MPPI_K1_WITH_DMA
  return 0;     // measured slower than the register ring: not instantiated in the product build
#endif
  static const int env = [] { const char* e = getenv("MPPI_K1_DMA"); return e ? atoi(e) : -1; }();
  if (env == 0) return 0;
  const long long nwg = (long long)((a.K + K1_BLOCK - 1) / K1_BLOCK) * a.n_env;
  int rows = env > 0 ? env : (nwg <= 256 ? 30 : 15);
  rows = rows >= 30 ? 30 : 15;
  auto fits = [&](int r) {
    return smem_tables + 16 + (size_t)(K1_BLOCK / WAVE) * (r == 30 ? DmaRing<NU, 30>::FLOATS_PER_WAVE : DmaRing<NU, 15>::FLOATS_PER_WAVE) * 4 <= 160 * 1024;
  };
  if (rows == 30 && !fits(30)) rows = 15;
  if (rows == 15 && !fits(15)) rows = 0;
  return rows;
}

template <class Model, typename T>
static int launch_rollout(const KArgs<T>& a_in, hipStream_t st) {
  constexpr int NU = Model::NU;
  if (a_in.W != nullptr) return launch_rollout_kmppi<Model, T>(a_in, st);   // mppi_rollout_cost_kmppi
  {
    // whole-command request on the engine's own generator with no row array: the on-chip form (or "not this path")
    const int r = launch_rollout_onchip<Model, T>(a_in, st);
    if (r != -1) return r;
  }
  {
    // a traced model with dense layers, fp32: sixteen samples per wave, the layers on the matrix cores (mlp_wide.hpp)
    const int r = launch_rollout_wide<Model, T>(a_in, st);
    if (r != -1) return r;
  }
  KArgs<T> a = a_in;
  // heavy models (one rolled copy of the step): rows in the engine's layout only (the caller converts a (K,T,nu) draw
  // on MPPI_E_UNSUPPORTED), one state rollout per action sequence
  if (model_heavy<Model>::value && (a.noise_src == MPPI_NOISE_KTN || a.M > 1)) return MPPI_E_UNSUPPORTED;
  const bool diag = a.diag != 0 || a.coloured != 0;   // a coloured stream runs the diagonal instantiation
  size_t smem = (size_t)(3 * a.J + BLOCK / WAVE + (a.diag != 0 ? 0 : 2 * NU * NU)) * sizeof(T);   // factors / Sigma^-1 of a coloured stream
  if (a.noise_src == MPPI_NOISE_KTN) {
    if (!(Ktn<NU>::OK && sizeof(T) == 4 && diag)) return MPPI_E_UNSUPPORTED;
    smem += (size_t)(4 + (K1_BLOCK / WAVE) * Ktn<NU>::LDS_FLOATS_PER_WAVE) * sizeof(T);
  }
  if (smem > 160 * 1024) return MPPI_E_UNSUPPORTED;   // T*nu beyond the LDS tables: not built
  const int dma = dma_rows_for<Model, T>(a, smem);
  if (dma == 30) smem += 16 + (size_t)(K1_BLOCK / WAVE) * DmaRing<NU, 30>::FLOATS_PER_WAVE * 4;
  else if (dma == 15) smem += 16 + (size_t)(K1_BLOCK / WAVE) * DmaRing<NU, 15>::FLOATS_PER_WAVE * 4;
  // Persistent grid: at most one workgroup per CU and environment-slice (the kernel holds one wave
  // per SIMD: 250+ registers), each walking its chunks blockIdx.x, +gridDim.x, ...  Launching all
  // K/256 workgroups instead lets the later ones start whenever a CU frees up, out of step with the
  // rest: measured 196 us against ~4 x 35 us at K = 262144.  MPPI_K1_PERSIST=0 restores the flat grid
  // (A/B runs), =n allows n workgroups per CU.
  const int nchunks = (a.K + K1_BLOCK - 1) / K1_BLOCK;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static const int persist = [] { const char* e = getenv("MPPI_K1_PERSIST"); return e ? atoi(e) : 1; }();
  int gx = nchunks;
  if (persist > 0) {
    const int cap = (n_cu * persist) / (a.n_env > 1 ? a.n_env : 1);
    if (gx > (cap > 1 ? cap : 1)) gx = cap > 1 ? cap : 1;
  }
  // Whole command in this one launch?  (the caller asked for it -- a.fuse >= 0 -- and left omega /
  // cost_total_non_zero NULL; small diagonal-form problem, rows in memory after K1, no sampler rows)
  const bool fuse = a.fuse >= 0 && a.M == 1 && a.ticket != nullptr && dma == 0 && diag && a.n_env == 1 && a.n_sampler == 0 &&
                    a.states == nullptr && a.omega == nullptr && a.wnz == nullptr && a.record != nullptr &&
                    (a.fuse == 0 || a.U_out != nullptr) && nchunks <= 64 && a.R == 1 && a.nkc == nchunks &&
                    a.Jpad <= 4 * UPD_TJ && a.z != nullptr &&
                    (a.noise_src == MPPI_NOISE_TNK4 || a.noise_src == MPPI_NOISE_PHILOX);
  if (fuse) gx = nchunks;
  const dim3 grid(gx, 1, a.n_env), block(K1_BLOCK);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  profile_next_events(&ev0, &ev1, &a.tstamp);   // null events == plain launch
  // event-attached launch only while measuring; the plain launch is what hipGraph capture records
#define MPPI_LAUNCH1(KERNEL)                                                                       \
  do {                                                                                             \
    if (smem > 64 * 1024) /* long horizons / DMA ring: the opt-in limit (160 KiB/CU) */            \
      (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)smem);                                                        \
    if (ev1 != nullptr) hipExtLaunchKernelGGL(KERNEL, grid, block, smem, st, ev0, ev1, 0, a);      \
    else hipLaunchKernelGGL(KERNEL, grid, block, smem, st, a);                                     \
  } while (0)
#define MPPI_LAUNCH(NOISE_)                                                                        \
  do {                                                                                             \
    if (diag) MPPI_LAUNCH1((rollout_cost_kernel<Model, T, NOISE_, true>));                         \
    else MPPI_LAUNCH1((rollout_cost_kernel<Model, T, NOISE_, false>));                             \
  } while (0)
#ifdef MPPI_K1_WITH_DMA
#define MPPI_K1_DMA_OK(ROWS_) (sizeof(T) == 4 && DmaRing<NU, ROWS_>::OK)
#else
#define MPPI_K1_DMA_OK(ROWS_) false
#endif
#define MPPI_LAUNCH_DMA(NOISE_, ROWS_)                                                             \
  do {                                                                                             \
    if constexpr (MPPI_K1_DMA_OK(ROWS_)) {                                                         \
      if (diag) MPPI_LAUNCH1((rollout_cost_kernel<Model, T, NOISE_, true, ROWS_>));                \
      else MPPI_LAUNCH1((rollout_cost_kernel<Model, T, NOISE_, false, ROWS_>));                    \
    }                                                                                              \
  } while (0)
  if (a.M > 1) {
    // several state rollouts per action sequence: rows in memory (the caller fills / converts them: standard normals, or KMPPI's
    // interpolated raw actions); MPPI, SMPPI (base sequence + smoothness cost) and KMPPI's two-launch form
    if (a.M > 4 || (a.noise_src != MPPI_NOISE_TNK4 && a.noise_src != MPPI_NOISE_ACTIONS)) return MPPI_E_UNSUPPORTED;
    if constexpr (!model_heavy<Model>::value) {
      // one wave per rollout copy where that form applies (rollout_copies.hpp): the same results, four times the waves
      const int rc = launch_rollout_copies<Model, T>(a_in, st);
      if (rc != -1) return rc;
      if (diag) MPPI_LAUNCH1((rollout_cost_kernel<Model, T, MPPI_NOISE_TNK4, true, 0, false, 4>));
      else MPPI_LAUNCH1((rollout_cost_kernel<Model, T, MPPI_NOISE_TNK4, false, 0, false, 4>));
    }
    return (int)hipGetLastError();
  }
  if (fuse) {
    if (a.noise_src == MPPI_NOISE_PHILOX) MPPI_LAUNCH1((rollout_cost_kernel<Model, T, MPPI_NOISE_PHILOX, true, 0, true>));
    else MPPI_LAUNCH1((rollout_cost_kernel<Model, T, MPPI_NOISE_TNK4, true, 0, true>));
    const int e = (int)hipGetLastError();
    return e != 0 ? e : MPPI_OK_FUSED;
  }
  if (a.noise_src == MPPI_NOISE_PHILOX) MPPI_LAUNCH(MPPI_NOISE_PHILOX);
  else if (a.noise_src == MPPI_NOISE_ACTIONS) {
    if (dma == 30) MPPI_LAUNCH_DMA(MPPI_NOISE_ACTIONS, 30);
    else if (dma == 15) MPPI_LAUNCH_DMA(MPPI_NOISE_ACTIONS, 15);
    else MPPI_LAUNCH(MPPI_NOISE_ACTIONS);
  } else if (a.noise_src == MPPI_NOISE_KTN) {
    if constexpr (Ktn<NU>::OK && sizeof(T) == 4 && !model_heavy<Model>::value) MPPI_LAUNCH1((rollout_cost_kernel<Model, T, MPPI_NOISE_KTN, true>));
  } else {
    if (dma == 30) MPPI_LAUNCH_DMA(MPPI_NOISE_TNK4, 30);
    else if (dma == 15) MPPI_LAUNCH_DMA(MPPI_NOISE_TNK4, 15);
    else MPPI_LAUNCH(MPPI_NOISE_TNK4);
  }
#undef MPPI_LAUNCH_DMA
#undef MPPI_LAUNCH
#undef MPPI_LAUNCH1
  return (int)hipGetLastError();
}

}  // namespace mppi
