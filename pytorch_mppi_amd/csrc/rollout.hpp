// rollout.hpp -- K1: the fused noise -> bound -> action-cost -> T-step rollout -> running-cost
// kernel.  One lane = one sample: the state lives in VGPRs for the whole horizon, the nominal
// sequence (shift applied on read) sits in LDS, the standard normals stream in sample-minor
// rows-of-4 (one 1 KiB coalesced read per wave instruction) or come from Philox in registers.
// Replaces mppi.py:407-417 (see include/mppi_amd.h).  Nothing of shape (K,T,nu) is written.
#pragma once
#include "actions.hpp"
#include "models.hpp"

namespace mppi {

// NOISE: MPPI_NOISE_TNK4 | _PHILOX | _ACTIONS (compile-time);  DIAG: diagonal Sigma
template <class Model, typename T, int NOISE, bool DIAG>
__global__ void __launch_bounds__(BLOCK) rollout_cost_kernel(const KArgs<T> a) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  constexpr bool SRC_ACTIONS = NOISE == MPPI_NOISE_ACTIONS;
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

  // Register ring of D super-steps of noise: with K = 65536 there is ONE wave per SIMD, so the
  // HBM latency (~1-2 us loaded) has to be covered by loads in flight from this wave alone:
  // ~24 rows-of-4 (384 B) per lane.
  constexpr int DWANT = (sizeof(T) == 4 ? 24 : 12) / P4;
  constexpr int D = DWANT < 2 ? 2 : (DWANT > 16 ? 16 : DWANT);
  const int nss = (a.Tn + TT - 1) / TT;
  auto fetch = [&](int ss, T (&dst)[P4 * 4]) {
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      T r[4];
      noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
      dst[4 * i + 0] = r[0]; dst[4 * i + 1] = r[1]; dst[4 * i + 2] = r[2]; dst[4 * i + 3] = r[3];
    }
  };
  T ring[D][P4 * 4];
  if constexpr (NOISE != MPPI_NOISE_PHILOX) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < nss) fetch(d, ring[d]);
  }

  T rollout = T(0), pert = T(0);
  for (int ss0 = 0; ss0 < nss; ss0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int ss = ss0 + d;
      if (ss < nss) {   // wave-uniform
        T zc[P4 * 4];
        if constexpr (NOISE == MPPI_NOISE_PHILOX) {
          fetch(ss, zc);
        } else {
#pragma unroll
          for (int i = 0; i < P4 * 4; ++i) zc[i] = ring[d][i];
          if (ss + D < nss) fetch(ss + D, ring[d]);   // refill this slot, D super-steps ahead
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int t = ss * TT + tt;
          if (t < a.Tn) {
            T z[NU], v[NU], e[NU], u[NU];
#pragma unroll
            for (int n = 0; n < NU; ++n) z[n] = zc[tt * NU + n];
            const T* srow = orow >= 0 ? a.sampler + ((long long)orow * a.Tn + t) * NU : nullptr;
            make_action<T, NU, DIAG, SRC_ACTIONS>(ac, Ue + t * NU, srow, z, orow, v, e);
            pert += action_cost_dot<T, NU, DIAG>(ac, Ue + t * NU, e);
#pragma unroll
            for (int n = 0; n < NU; ++n) u[n] = a.u_scale * v[n];          // mppi.py:313
            model.step(x, u, t);                                           // :314
            rollout += model.cost(x, u, t);                                // :318-319
            if (a.states != nullptr && active) {
              T* __restrict__ so = a.states + ((long long)k * a.Tn + t) * NX;
#pragma unroll
              for (int i = 0; i < NX; ++i) so[i] = x[i];                   // :321
            }
          }
        }
      }
    }
  }
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
