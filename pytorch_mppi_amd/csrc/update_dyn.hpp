// update_dyn.hpp -- the control width nu as a RUNTIME value.
//
// The compiled-in widths (MPPI_NU_LIST in update.hip) keep whole action vectors in registers.  A
// controller of any other width (9, 11, 20, ...) must still work -- the reference takes any
// noise_sigma -- so these kernels run the same pipeline (actions.hpp) with the per-sample vectors in
// per-thread LDS columns [n][DYN_BLOCK] and runtime loops over n.  They are the general fallback of
//   mppi_prepare          (generic path, lazy noise / perturbed_action)      mppi.py:375-400, :186-199
//   mppi_weights_partial  (full Sigma; the diagonal K3 is width-agnostic)    :254-275
//   mppi_kmppi_interp     (KMPPI)                                            :621-670
// Same expression order as make_action / action_cost_dot, so results match the templated kernels to
// rounding.  Correctness first: one element per memory instruction, no attempt at the roofline.
#pragma once
#include "actions.hpp"

namespace mppi {

constexpr int DYN_BLOCK = 64;

// element j of the sample-minor rows-of-4 stream of sample k, through a one-row cache
template <typename T, int NOISE>
struct RowCursor {
  long long row = -1;
  T r[4];
  __device__ __forceinline__ T get(const KArgs<T>& a, int k, long long j, bool store_generated) {
    const long long jb = j >> 2;
    if (jb != row) {
      row = jb;
      noise4<T, NOISE>(a, jb, k, r);
      if constexpr (NOISE == MPPI_NOISE_PHILOX) {
        if (store_generated && a.z != nullptr) store4<T>(const_cast<T*>(a.z), a.zp, jb, k, r);
      }
    }
    const int c = (int)(j & 3);
    return c == 0 ? r[0] : c == 1 ? r[1] : c == 2 ? r[2] : r[3];
  }
};

// zc -> vc (bounded action), ec (post-clamp noise); columns are [n * DYN_BLOCK + threadIdx.x]
template <typename T>
__device__ __forceinline__ void make_action_dyn(const KArgs<T>& a, const T* __restrict__ Ut,
                                                const T* __restrict__ srow, int orow, const T* zc,
                                                T* vc, T* ec) {
  const int nu = a.nu, tid = threadIdx.x;
  for (int n = 0; n < nu; ++n) {
    T v;
    if (a.noise_src == MPPI_NOISE_ACTIONS) {
      v = zc[n * DYN_BLOCK + tid];
    } else if (a.coloured) {
      v = Ut[n] + zc[n * DYN_BLOCK + tid];
    } else if (a.diag) {
      v = Ut[n] + (zc[n * DYN_BLOCK + tid] * a.L[n * nu + n] + a.mu[n]);
    } else {
      T s = zc[tid] * a.L[n * nu];
      for (int m = 1; m <= n; ++m) s += zc[m * DYN_BLOCK + tid] * a.L[n * nu + m];   // lower triangular
      v = Ut[n] + (s + a.mu[n]);
    }
    if (orow == -1) v = T(0);
    else if (orow >= 0) v = srow[n];
    v = clampT(v, a.umin[n], a.umax[n]);
    vc[n * DYN_BLOCK + tid] = v;
    ec[n * DYN_BLOCK + tid] = (v - Ut[n]) * a.e_scale;
  }
}

template <typename T>
__device__ __forceinline__ T action_cost_dyn(const KArgs<T>& a, const T* __restrict__ Un, const T* ec) {
  const int nu = a.nu, tid = threadIdx.x;
  T acc = T(0);
  for (int n = 0; n < nu; ++n) {
    T p;
    if (a.diag) {
      const T e = ec[n * DYN_BLOCK + tid];
      p = Un[n] * ((a.lambda_ * (a.abs_cost ? m_abs(e) : e)) * a.sinv[n * nu + n]);
    } else {
      T s = T(0);
      for (int m = 0; m < nu; ++m) {
        const T e = ec[m * DYN_BLOCK + tid];
        const T q = (a.lambda_ * (a.abs_cost ? m_abs(e) : e)) * a.sinv[m * nu + n];
        s = (m == 0) ? q : s + q;
      }
      p = Un[n] * s;
    }
    acc = (n == 0) ? p : acc + p;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// prepare (mppi_prepare): perturbed_action / noise (K,T,nu) + pert_cost (K)
// LDS: Ue[J] | Un[J] | zc, vc, ec, vprev [nu][DYN_BLOCK]
// ---------------------------------------------------------------------------------------------
template <typename T, int NOISE>
__global__ void __launch_bounds__(DYN_BLOCK) prepare_dyn_kernel(const KArgs<T> a_in) {
  const KArgs<T> a = env_view(a_in);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);
  T* Un = Ue + a.J;
  T* zc = Un + a.J;
  T* vc = zc + a.nu * DYN_BLOCK;
  T* ec = vc + a.nu * DYN_BLOCK;
  T* vp = ec + a.nu * DYN_BLOCK;
  for (int j = threadIdx.x; j < a.J; j += DYN_BLOCK) { Ue[j] = u_base(a, j); Un[j] = u_eff(a, j); }
  __syncthreads();
  const int k = blockIdx.x * DYN_BLOCK + threadIdx.x;
  if (k >= a.K) return;
  const int nu = a.nu, tid = threadIdx.x;
  const int orow = overwrite_row(a, a.k_offset + k);
  RowCursor<T, NOISE> cur;
  T pert = T(0), smooth = T(0);
  for (int n = 0; n < nu; ++n) vp[n * DYN_BLOCK + tid] = T(0);
  for (int t = 0; t < a.Tn; ++t) {
    for (int n = 0; n < nu; ++n) zc[n * DYN_BLOCK + tid] = cur.get(a, k, (long long)t * nu + n, true);
    const T* srow = orow >= 0 ? a.sampler + ((long long)orow * a.Tn + t) * nu : nullptr;
    make_action_dyn<T>(a, Ue + t * nu, srow, orow, zc, vc, ec);
    pert += action_cost_dyn<T>(a, Un + t * nu, ec);
    if (a.smooth_w != T(0)) {                           // mppi.py:559-562
      T d2 = T(0);
      for (int n = 0; n < nu; ++n) {
        const T d = vc[n * DYN_BLOCK + tid] - vp[n * DYN_BLOCK + tid];
        d2 += d * d;
        vp[n * DYN_BLOCK + tid] = vc[n * DYN_BLOCK + tid];
      }
      if (t > 0) smooth += d2;
    }
    const long long o = ((long long)k * a.Tn + t) * nu;
    for (int n = 0; n < nu; ++n) {
      if (a.pa != nullptr) a.pa[o + n] = vc[n * DYN_BLOCK + tid];
      if (a.noise != nullptr) a.noise[o + n] = ec[n * DYN_BLOCK + tid];
    }
  }
  // generate-once Philox: the padded tail rows of the stream must exist for K3's whole-row reads
  if constexpr (NOISE == MPPI_NOISE_PHILOX) {
    if (a.z != nullptr && (a.J & 3) != 0) (void)cur.get(a, k, (long long)a.J - 1, true);
  }
  if (a.pert != nullptr) a.pert[k] = pert + a.smooth_w * smooth;
}

// ---------------------------------------------------------------------------------------------
// K3, full Sigma: block (kc, t) = samples [kc*R*DYN_BLOCK, +R*DYN_BLOCK) x the nu columns of
// timestep t.  Each thread accumulates w*eps' for its samples in an LDS column; the columns are
// then summed over the 64 threads in fixed order.  nkc here = ceil(K / (R*DYN_BLOCK)) <= carve's.
// LDS: Ue[J] | zc, vc, ec, acc [nu][DYN_BLOCK]
// ---------------------------------------------------------------------------------------------
template <typename T, int NOISE>
__global__ void __launch_bounds__(DYN_BLOCK) weights_partial_full_dyn_kernel(const KArgs<T> a_in, int per_block) {
  const KArgs<T> a = env_view(a_in);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);
  T* zc = Ue + a.J;
  T* vc = zc + a.nu * DYN_BLOCK;
  T* ec = vc + a.nu * DYN_BLOCK;
  T* ac = ec + a.nu * DYN_BLOCK;
  const int nu = a.nu, tid = threadIdx.x;
  const int kc = blockIdx.x, t = blockIdx.y;
  for (int j = tid; j < a.J; j += DYN_BLOCK) Ue[j] = u_base(a, j);
  // beta of the shard: fixed-order minimum over the per-wave minima (one wave here)
  T bm = inf_v<T>();
  for (int i = tid; i < a.nb1; i += DYN_BLOCK) { const T v = a.block_min[i]; bm = v < bm ? v : bm; }
  const T beta = wave_min<T>(bm);
  const T inv_lambda = T(1) / a.lambda_;
  for (int n = 0; n < nu; ++n) ac[n * DYN_BLOCK + tid] = T(0);
  __syncthreads();
  T eta = T(0);
  for (int i = 0; i < per_block; i += DYN_BLOCK) {
    const int k = kc * per_block + i + tid;
    const bool ok = k < a.K;
    const int kq = ok ? k : a.K - 1;
    const T w = ok ? weight_of<T>(a.cost[kq], beta, inv_lambda) : T(0);
    const int orow = ok ? overwrite_row(a, a.k_offset + k) : -2;
    eta += w;
    if (ok && t == 0 && a.wnz != nullptr) a.wnz[k] = w;
    RowCursor<T, NOISE> cur;
    for (int n = 0; n < nu; ++n) zc[n * DYN_BLOCK + tid] = cur.get(a, kq, (long long)t * nu + n, false);
    const T* srow = orow >= 0 ? a.sampler + ((long long)orow * a.Tn + t) * nu : nullptr;
    make_action_dyn<T>(a, Ue + t * nu, srow, orow, zc, vc, ec);
    for (int n = 0; n < nu; ++n) ac[n * DYN_BLOCK + tid] += w * ec[n * DYN_BLOCK + tid];
  }
  __syncthreads();
  for (int n = tid; n < nu; n += DYN_BLOCK) {
    T s = ac[n * DYN_BLOCK];
    for (int q = 1; q < DYN_BLOCK; ++q) s += ac[n * DYN_BLOCK + q];
    a.P_part[(long long)kc * a.Jpad + t * nu + n] = s;
  }
  const T eta_b = wave_sum<T>(eta);
  if (t == 0 && tid == 0) a.eta_part[kc] = eta_b;
}

// ---------------------------------------------------------------------------------------------
// KMPPI interpolation: v_raw[t] = sum_s W[t,s] * clamp(theta[s] + colour(z_S)[s])
// LDS: ctrl[S*nu][DYN_BLOCK] | zc, vc, ec [nu][DYN_BLOCK].  `a` describes the support-point stream.
// ---------------------------------------------------------------------------------------------
template <typename T, int NOISE>
__global__ void __launch_bounds__(DYN_BLOCK) kmppi_interp_dyn_kernel(const KArgs<T> a, const T* __restrict__ W,
                                                                    int Thor, int J4out, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ctrl = reinterpret_cast<T*>(smem_raw);
  T* zc = ctrl + (size_t)a.J * DYN_BLOCK;
  T* vc = zc + a.nu * DYN_BLOCK;
  T* ec = vc + a.nu * DYN_BLOCK;
  const int nu = a.nu, tid = threadIdx.x, S = a.Tn;
  const int kraw = blockIdx.x * DYN_BLOCK + tid;
  const bool active = kraw < a.K;
  const int k = active ? kraw : a.K - 1;
  RowCursor<T, NOISE> cur;
  for (int s = 0; s < S; ++s) {
    for (int n = 0; n < nu; ++n) zc[n * DYN_BLOCK + tid] = cur.get(a, k, (long long)s * nu + n, false);
    make_action_dyn<T>(a, a.U + s * nu, nullptr, -2, zc, vc, ec);      // theta + eps, clamp (mppi.py:660-663)
    for (int n = 0; n < nu; ++n) ctrl[(s * nu + n) * DYN_BLOCK + tid] = vc[n * DYN_BLOCK + tid];
  }
  if (!active) return;
  const long long Jout = (long long)Thor * nu;
  for (long long j = 0; j < (long long)J4out * 4; ++j) {
    T acc = T(0);
    if (j < Jout) {
      const int t = (int)(j / nu), n = (int)(j - (long long)t * nu);
      for (int s = 0; s < S; ++s) acc += W[t * S + s] * ctrl[(s * nu + n) * DYN_BLOCK + tid];   // :665
    }
    out[((j >> 2) * a.zp + k) * 4 + (j & 3)] = acc;        // padded tail = 0
  }
}

}  // namespace mppi
