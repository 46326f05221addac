// weights.hpp -- device pieces of the exp-weighted update (K3) shared by the stand-alone K3 kernels
// (update.hip) and the single-launch small-problem command (rollout.hpp): the shard minimum, the
// weight function of mppi.py:12-13 / :256, the streaming tile loop of the diagonal K3 and the
// correction for overwritten rows.
#pragma once
#include "actions.hpp"

namespace mppi {

template <typename T>
__device__ __forceinline__ T shard_beta(const KArgs<T>& a, T* red) {
  T m = inf_v<T>();
  for (int i = threadIdx.x; i < a.nb1; i += BLOCK) {
    const T v = a.block_min[i];
    m = v < m ? v : m;
  }
  return block_min<T>(m, red);
}

// mppi.py:12-13 / :256: exp(-(1/lambda) * (cost - beta))
template <typename T>
__device__ __forceinline__ T weight_of(T cost, T beta, T inv_lambda) {
  return m_exp(-inv_lambda * (cost - beta));
}

// Overwritten rows (sample_null_action / sampler rows, mppi.py:387-400) are masked out of the
// streaming loop (weight 0) and added back afterwards, one lane per column: their "noise" is
// clamp(0 | sampler action) - U, independent of z.  Returns the correction for column j.
template <typename T>
__device__ __forceinline__ T overwrite_correction(const KArgs<T>& a, int kbeg, int kend, int j,
                                                  T uj, T lo, T hi, T beta, T inv_lambda) {
  // rows are global indices [0, n_over); this block covers local samples [kbeg, kend)
  const long long n_over = (a.null_action ? 1 : 0) + (long long)a.n_sampler;
  T corr = T(0);
  for (int k = kbeg; k < kend; ++k) {
    const long long kg = a.k_offset + k;
    if (kg >= n_over) break;
    const int orow = overwrite_row(a, kg);
    T v = T(0);
    if (orow >= 0) v = a.sampler[(long long)orow * a.J + j];
    v = clampT(v, lo, hi);
    corr += weight_of<T>(a.cost[k], beta, inv_lambda) * (v - uj);
  }
  return corr;
}

// the streaming loop of the diagonal K3 over one 64-column tile; SPARSE: 64-sample groups whose
// weights are all exactly zero (live[r] == false, wave-uniform) are skipped -- no load, no RNG
template <typename T, int NOISE, int R, bool SPARSE>
__device__ __forceinline__ void k3_tile_loop(const KArgs<T>& a, int jt, int nrows, const int (&kk)[R],
                                             const T (&w)[R], const bool (&live)[R], const T* cU,
                                             const T* cS, const T* cM, const T* cLo, const T* cHi,
                                             T (&acc)[UPD_TJ]) {
#pragma unroll
  for (int jbl = 0; jbl < UPD_TJ / 4; ++jbl) {
    if (jbl < nrows) {
      const long long jb = (long long)jt * (UPD_TJ / 4) + jbl;
      T zz[R][4];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!SPARSE || live[r]) noise4_last<T, NOISE>(a, jb, kk[r], zz[r]);
      }
      T u4[4], s4[4], m4[4], lo4[4], hi4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u4[c] = cU[4 * jbl + c]; s4[c] = cS[4 * jbl + c]; m4[c] = cM[4 * jbl + c];
        lo4[c] = cLo[4 * jbl + c]; hi4[c] = cHi[4 * jbl + c];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (!SPARSE || live[r]) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            T v = u4[c] + (zz[r][c] * s4[c] + m4[c]);
            v = clampT(v, lo4[c], hi4[c]);
            acc[4 * jbl + c] += w[r] * (v - u4[c]);
          }
        }
      }
    }
  }
}


}  // namespace mppi
