// rollout.hpp -- K1: the fused noise -> bound -> action-cost -> T-step rollout -> running-cost
// kernel.  One lane = one sample: the state lives in VGPRs for the whole horizon, the nominal
// sequence (shift applied on read) sits in LDS, the standard normals stream in sample-minor
// rows-of-4 (one 1 KiB coalesced read per wave instruction) or come from Philox in registers.
// Replaces mppi.py:407-417 (see include/mppi_amd.h).  Nothing of shape (K,T,nu) is written.
//
// Memory pipeline (the part that decides the HBM fraction): at K = 65536 the chip holds ONE wave
// per SIMD, so latency must be covered by loads in flight from that wave alone.  The noise rows
// go through a register ring of D super-steps (~24 rows-of-4 = 384 B per lane outstanding).
// gfx950 retires vector-memory ops in order behind ONE counter (vmcnt), and the compiler can
// only wait for "all but the N youngest": any load or store under a branch between two ring
// slots makes N unknowable and every wait collapses to vmcnt(0), which serialises the ring.
// Hence the main streaming loop contains NO branch and NO conditional vector-memory instruction:
// it runs whole groups of D complete super-steps, refills are unconditional (row index clamped),
// the horizon tail is consumed from the ring after the loop, and the rare work that needs extra
// memory traffic (sampler rows, `states` stores) lives in a separate instantiation (SLOW) chosen
// per wave.
#pragma once
#include "actions.hpp"
#include "models.hpp"

namespace mppi {

// one timestep: actions from z, action cost, dynamics, running cost
template <class Model, typename T, int NOISE, bool DIAG, bool SLOW>
__device__ __forceinline__ void rollout_step(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                             const Model& model, const T* __restrict__ Ue, int k,
                                             bool active, int orow, int t, const T* zt,
                                             T (&x)[Model::NX], T& rollout, T& pert) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr bool SRC_ACTIONS = NOISE == MPPI_NOISE_ACTIONS;
  T z[NU], v[NU], e[NU], u[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) z[n] = zt[n];
  const T* srow = nullptr;
  if constexpr (SLOW) srow = orow >= 0 ? a.sampler + ((long long)orow * a.Tn + t) * NU : nullptr;
  make_action<T, NU, DIAG, SRC_ACTIONS>(ac, Ue + t * NU, srow, z, SLOW ? orow : -2, v, e);
  pert += action_cost_dot<T, NU, DIAG>(ac, Ue + t * NU, e);
#pragma unroll
  for (int n = 0; n < NU; ++n) u[n] = a.u_scale * v[n];          // mppi.py:313
  model.step(x, u, t);                                           // :314
  rollout += model.cost(x, u, t);                                // :318-319
  if constexpr (SLOW) {
    if (a.states != nullptr && active) {
      T* __restrict__ so = a.states + ((long long)k * a.Tn + t) * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) so[i] = x[i];                 // :321
    }
  }
}

template <class Model, typename T, int NOISE, bool DIAG, bool SLOW>
__device__ __forceinline__ void rollout_stream(const KArgs<T>& a, const ActionConsts<T, Model::NU>& ac,
                                               const Model& model, const T* __restrict__ Ue, int k,
                                               bool active, int orow, T (&x)[Model::NX], T& rollout,
                                               T& pert) {
  constexpr int NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  const int nss = (a.Tn + TT - 1) / TT;

  if constexpr (NOISE == MPPI_NOISE_PHILOX) {
    // generated in registers: no memory pipeline to manage
    for (int ss = 0; ss < nss; ++ss) {
      T zc[P4 * 4];
#pragma unroll
      for (int i = 0; i < P4; ++i) {
        T r[4];
        noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
        zc[4 * i + 0] = r[0]; zc[4 * i + 1] = r[1]; zc[4 * i + 2] = r[2]; zc[4 * i + 3] = r[3];
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int t = ss * TT + tt;
        if (t < a.Tn)
          rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, Ue, k, active, orow, t, zc + tt * NU, x,
                                                    rollout, pert);
      }
    }
    return;
  } else {
    // ring depth: ~24 rows-of-4 in flight per lane, at most 16 timesteps ahead
    constexpr int DROWS = (sizeof(T) == 4 ? 24 : 12) / P4;
    constexpr int DSTEP = 16 / TT;
    constexpr int D0 = DROWS < DSTEP ? DROWS : DSTEP;
    constexpr int D = D0 < 2 ? 2 : D0;
    const int last = nss - 1;
    const int nss_full = a.Tn / TT;          // super-steps whose TT timesteps all exist
    T ring[D][P4 * 4];
    int kf = k;                              // sample index as seen by the refill loads
    auto fetch = [&](int ss, T (&dst)[P4 * 4]) {
#pragma unroll
      for (int i = 0; i < P4; ++i) {
        T r[4];
        noise4<T, NOISE>(a, (long long)ss * P4 + i, kf, r);
        dst[4 * i + 0] = r[0]; dst[4 * i + 1] = r[1]; dst[4 * i + 2] = r[2]; dst[4 * i + 3] = r[3];
      }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d < last ? d : last, ring[d]);

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
        for (int tt = 0; tt < TT; ++tt)
          rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, Ue, k, active, orow, ss * TT + tt,
                                                    &ring[d][tt * NU], x, rollout, pert);
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
            rollout_step<Model, T, NOISE, DIAG, SLOW>(a, ac, model, Ue, k, active, orow, t,
                                                      &ring[d][tt * NU], x, rollout, pert);
        }
      }
    }
  }
}

// NOISE: MPPI_NOISE_TNK4 | _PHILOX | _ACTIONS (compile-time);  DIAG: diagonal Sigma
template <class Model, typename T, int NOISE, bool DIAG>
__global__ void __launch_bounds__(BLOCK) rollout_cost_kernel(const KArgs<T> a) {
  constexpr int NX = Model::NX, NU = Model::NU;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);   // [J] nominal sequence, shift applied
  T* red = Ue + a.J;                        // [BLOCK/WAVE]
  T* fac = red + BLOCK / WAVE;              // [2*NU*NU] full-Sigma factors (only if !DIAG)

  ActionConsts<T, NU> ac;
  ac.load(a, DIAG ? nullptr : fac);
  for (int j = threadIdx.x; j < a.J; j += BLOCK) Ue[j] = u_eff(a, j);
  __syncthreads();

  const int kraw = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = kraw < a.K;
  const int k = active ? kraw : a.K - 1;   // tail lanes shadow the last sample, never store
  const long long kg = a.k_offset + k;
  const int orow = overwrite_row(a, kg);

  const Model model(a);
  T x[NX];
  {
    const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = s0[i];      // mppi.py:302-305
  }

  T rollout = T(0), pert = T(0);
  // wave-uniform choice: does this wave own overwritten rows, or must it store the states?
  const bool slow = __any(orow != -2) || a.states != nullptr;
  if (slow)
    rollout_stream<Model, T, NOISE, DIAG, true>(a, ac, model, Ue, k, active, orow, x, rollout, pert);
  else
    rollout_stream<Model, T, NOISE, DIAG, false>(a, ac, model, Ue, k, active, orow, x, rollout, pert);

  if (a.use_terminal) rollout += model.terminal(x);                    // :324-328
  const T total = rollout + pert;                                      // :416
  if (active) {
    a.cost[k] = total;
    if (a.pert != nullptr) a.pert[k] = pert;
  }
  const T bm = block_min<T>(active ? total : inf_v<T>(), red);
  if (threadIdx.x == 0) a.block_min[blockIdx.x] = bm;
}

template <class Model, typename T>
static int launch_rollout(const KArgs<T>& a, hipStream_t st) {
  constexpr int NU = Model::NU;
  const bool diag = a.diag != 0;
  const size_t smem = (size_t)(a.J + BLOCK / WAVE + (diag ? 0 : 2 * NU * NU)) * sizeof(T);
  const dim3 grid((a.K + BLOCK - 1) / BLOCK), block(BLOCK);
#define MPPI_LAUNCH(NOISE_)                                                                        \
  do {                                                                                             \
    if (diag)                                                                                      \
      hipLaunchKernelGGL((rollout_cost_kernel<Model, T, NOISE_, true>), grid, block, smem, st, a); \
    else                                                                                           \
      hipLaunchKernelGGL((rollout_cost_kernel<Model, T, NOISE_, false>), grid, block, smem, st, a);\
  } while (0)
  if (a.noise_src == MPPI_NOISE_PHILOX) MPPI_LAUNCH(MPPI_NOISE_PHILOX);
  else if (a.noise_src == MPPI_NOISE_ACTIONS) MPPI_LAUNCH(MPPI_NOISE_ACTIONS);
  else MPPI_LAUNCH(MPPI_NOISE_TNK4);
#undef MPPI_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace mppi
