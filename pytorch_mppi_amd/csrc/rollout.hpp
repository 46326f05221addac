// rollout.hpp -- K1: the fused noise -> bound -> action-cost -> T-step rollout -> running-cost
// kernel.  One lane = one sample: the state lives in VGPRs for the whole horizon, the nominal
// sequence (shift applied on read) sits in LDS, the standard normals stream in sample-minor
// rows-of-4 (one 1 KiB coalesced read per wave instruction) or come from Philox in registers.
// Replaces mppi.py:407-417 (see include/mppi_amd.h).  Nothing of shape (K,T,nu) is written.
#pragma once
#include "actions.hpp"
#include "models.hpp"

namespace mppi {

template <class Model, typename T, int NOISE>
__global__ void __launch_bounds__(BLOCK) rollout_cost_kernel(const KArgs<T> a) {
  constexpr int NX = Model::NX, NU = Model::NU;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);   // [J] nominal sequence, shift applied
  T* red = Ue + a.J;                        // [BLOCK/WAVE]

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

  T zc[P4 * 4], zn[P4 * 4];
  const int nss = (a.Tn + TT - 1) / TT;
  auto fetch = [&](int ss, T (&dst)[P4 * 4]) {
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      T r[4];
      noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
      dst[4 * i + 0] = r[0]; dst[4 * i + 1] = r[1]; dst[4 * i + 2] = r[2]; dst[4 * i + 3] = r[3];
    }
  };
  if constexpr (NOISE != MPPI_NOISE_PHILOX) fetch(0, zn);

  T rollout = T(0), pert = T(0);
  for (int ss = 0; ss < nss; ++ss) {
    if constexpr (NOISE == MPPI_NOISE_PHILOX) {
      fetch(ss, zc);
    } else {
#pragma unroll
      for (int i = 0; i < P4 * 4; ++i) zc[i] = zn[i];
      fetch(ss + 1 < nss ? ss + 1 : ss, zn);   // software prefetch of the next super-step
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = ss * TT + tt;
      if (t < a.Tn) {
        T z[NU], v[NU], e[NU], u[NU];
#pragma unroll
        for (int n = 0; n < NU; ++n) z[n] = zc[tt * NU + n];
        make_action<T, NU>(a, Ue, t, z, orow, v, e);
        pert += action_cost_dot<T, NU>(a, Ue, t, e);
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
  const size_t smem = (size_t)(a.J + BLOCK / WAVE) * sizeof(T);
  const dim3 grid((a.K + BLOCK - 1) / BLOCK), block(BLOCK);
  if (a.noise_src == MPPI_NOISE_PHILOX)
    hipLaunchKernelGGL((rollout_cost_kernel<Model, T, MPPI_NOISE_PHILOX>), grid, block, smem, st, a);
  else
    hipLaunchKernelGGL((rollout_cost_kernel<Model, T, MPPI_NOISE_TNK4>), grid, block, smem, st, a);
  return (int)hipGetLastError();
}

}  // namespace mppi
