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


// The diagonal K3 of ONE workgroup (kc, jt): samples [kc*BLOCK*R, +BLOCK*R), columns [jt*64, +64) -- the body of
// weights_partial_diag_kernel (update.hip), shared with the launch that carries the NEXT command's draw in workgroups of its
// own beside K3's (noise_torch.hip weights_partial_rows_kernel).  BLOCK threads; the environment is blockIdx.z.
template <typename T, int NOISE, int R>
__device__ __forceinline__ void k3_diag_block(const KArgs<T>& a_in, const int kc, const int jt) {
  const KArgs<T> a = env_view(a_in);
  __shared__ __attribute__((aligned(16))) T cU[UPD_TJ], cS[UPD_TJ], cM[UPD_TJ], cLo[UPD_TJ], cHi[UPD_TJ];
  __shared__ T red[BLOCK / WAVE];
  __shared__ T wsum[BLOCK / WAVE][UPD_TJ];
  const int j0 = jt * UPD_TJ;
  if (threadIdx.x < UPD_TJ) {
    const int j = j0 + threadIdx.x;
    const bool ok = j < a.J;
    const int n = ok ? j % a.nu : 0;
    cU[threadIdx.x] = ok ? u_base(a, j) : T(0);
    cS[threadIdx.x] = ok ? (a.coloured ? T(1) : a.L[n * a.nu + n]) : T(0);     // coloured stream: eps is in z
    cM[threadIdx.x] = (ok && !a.coloured) ? a.mu[n] : T(0);
    cLo[threadIdx.x] = ok ? a.umin[n] : T(0);
    cHi[threadIdx.x] = ok ? a.umax[n] : T(0);
  }
  const T beta = shard_beta(a, red);   // contains the barriers that publish the constants
  const T inv_lambda = T(1) / a.lambda_;
  const long long n_over = (a.null_action ? 1 : 0) + (long long)a.n_sampler;

  T w[R];
  int kk[R];
  T eta = T(0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = (kc * R + r) * BLOCK + threadIdx.x;
    const bool ok = k < a.K;
    kk[r] = ok ? k : a.K - 1;
    const T wr = ok ? weight_of<T>(a.cost[kk[r]], beta, inv_lambda) : T(0);
    eta += wr;
    if (ok && jt == 0 && a.wnz != nullptr) a.wnz[k] = wr;
    w[r] = (a.k_offset + k < n_over) ? T(0) : wr;   // overwritten rows: see overwrite_correction
  }

  T acc[UPD_TJ];
#pragma unroll
  for (int i = 0; i < UPD_TJ; ++i) acc[i] = T(0);

  // Samples whose weight is EXACTLY zero (exp underflow: (cost - beta)/lambda > ~104 in fp32) add
  // exactly nothing to any column, so their rows are neither read nor generated.  With the peaked
  // softmax of everyday MPPI settings (N_eff of tens to hundreds among 65536) that is almost every
  // 64-sample group; with a flat softmax every group is live and the dense loop runs unchanged.
  bool live[R], all_live = true, any_live = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    live[r] = __ballot(w[r] != T(0)) != 0ull;          // wave-uniform
    all_live = all_live && live[r];
    any_live = any_live || live[r];
  }
  const int nrows = a.J4 - jt * (UPD_TJ / 4);   // rows-of-4 of this tile that exist (block-uniform)
  if (all_live)
    k3_tile_loop<T, NOISE, R, false>(a, jt, nrows, kk, w, live, cU, cS, cM, cLo, cHi, acc);
  else if (any_live)
    k3_tile_loop<T, NOISE, R, true>(a, jt, nrows, kk, w, live, cU, cS, cM, cLo, cHi, acc);

  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  // lane l: this wave's sum of column j0+l (all accumulators are still zero without a live group)
  const T colsum = any_live ? wave_reduce_transpose64<T>(acc) : T(0);
  wsum[wv][lane] = colsum;
  const T eta_b = block_sum<T>(eta, red);             // barriers also publish wsum
  if (threadIdx.x < UPD_TJ) {
    T s = wsum[0][threadIdx.x];
#pragma unroll
    for (int i = 1; i < BLOCK / WAVE; ++i) s += wsum[i][threadIdx.x];
    const int j = j0 + threadIdx.x;
    const int kbeg = kc * R * BLOCK;
    if (a.k_offset + kbeg < n_over && j < a.J) {
      const int kend = (kbeg + R * BLOCK) < a.K ? (kbeg + R * BLOCK) : a.K;
      s += overwrite_correction<T>(a, kbeg, kend, j, cU[threadIdx.x], cLo[threadIdx.x], cHi[threadIdx.x],
                                   beta, inv_lambda);
    }
    if (j < a.Jpad) a.P_part[(long long)kc * a.Jpad + j] = s * a.e_scale;   // 1 | 1/dt (SMPPI)
  }
  if (jt == 0 && threadIdx.x == 0) a.eta_part[kc] = eta_b;
}

}  // namespace mppi
