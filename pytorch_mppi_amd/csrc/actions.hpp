// actions.hpp -- per-sample, per-timestep action pipeline shared by K1 (fused rollout),
// mppi_prepare (generic path / lazy attributes), kmppi_interp and the full-Sigma K3:
//   eps  = z*sqrt(diag)+mu | L z + mu                         mppi.py:201-206
//   v    = U[t] + eps                                          :380
//   v    = 0 (global row 0, sample_null_action) | sampler row  :387-400
//   v    = clamp(v, u_min, u_max)                              :383, :419-420
//   e    = v - U[t]              (post-clamp noise)            :385
//   a    = lambda*e*diag^-1 | (lambda*e) Sigma^-1 (|e| if abs) :186-199
//   pert += sum_n U[t,n]*a[n]                                  :415
// The per-control constants are read ONCE at kernel entry (before any store, so the backend can
// use scalar loads and keep them in SGPRs); full-Sigma factors are staged in LDS.
#pragma once
#include "common.hpp"

namespace mppi {

// nominal sequence with shift_nominal_trajectory (mppi.py:232-238) applied on read
template <typename T>
__device__ __forceinline__ T u_eff(const KArgs<T>& a, int j) {
  if (!a.shift) return a.U[j];
  const int jn = j + a.nu;
  return jn < a.J ? a.U[jn] : a.u_init[jn - a.J];
}

// the sequence the noise is added to and measured from: the (shifted) nominal U, or -- SMPPI --
// the host-provided base A + U*dt (mppi.py:540)
template <typename T>
__device__ __forceinline__ T u_base(const KArgs<T>& a, int j) {
  return a.B != nullptr ? a.B[j] : u_eff(a, j);
}

// which overwrite row (if any) global sample kg is: -2 none, -1 null action, >=0 sampler row
template <typename T>
__device__ __forceinline__ int overwrite_row(const KArgs<T>& a, long long kg) {
  if (a.null_action && kg == 0) return -1;
  const long long r = kg - (a.null_action ? 1 : 0);
  if (r >= 0 && r < a.n_sampler) return (int)r;
  return -2;
}

template <typename T, int NU>
struct ActionConsts {
  T sd[NU], mu[NU], lo[NU], hi[NU], ci[NU];   // sqrt(diag), mu, bounds, 1/diag
  const T* Lm;                                // (NU,NU) chol(Sigma), row-major   (full Sigma only)
  const T* Sm;                                // (NU,NU) Sigma^-1
  T lambda_, e_scale;
  int abs_cost;
  // `lds` may be nullptr when Sigma is diagonal.  Caller must __syncthreads() afterwards.
  __device__ __forceinline__ void load(const KArgs<T>& a, T* lds) {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      sd[n] = a.coloured ? T(1) : a.L[n * NU + n];     // coloured stream: eps is in z already
      mu[n] = a.coloured ? T(0) : a.mu[n];
      lo[n] = a.umin[n];
      hi[n] = a.umax[n];
      ci[n] = a.sinv[n * NU + n];
    }
    lambda_ = a.lambda_;
    e_scale = a.e_scale;
    abs_cost = a.abs_cost;
    Lm = lds;
    Sm = lds + NU * NU;
    if (!a.diag && lds != nullptr) {
      for (int i = threadIdx.x; i < NU * NU; i += blockDim.x) {
        lds[i] = a.L[i];
        lds[NU * NU + i] = a.sinv[i];
      }
    }
  }
};

// row n of chol(Sigma) (lower triangular: entries 0..n) out of the LDS copy.  One uniform-address
// ds_read_b128 per 16 bytes when the rows are 16-byte aligned (NU*sizeof(T) % 16 == 0 and the factor
// block starts on a 16-byte boundary, which every kernel's LDS carve guarantees for such NU):
// the full-Sigma kernels are LDS-issue-bound on these reads, not FMA-bound.
template <typename T, int NU>
__device__ __forceinline__ void chol_row(const T* __restrict__ Lm, int n, T (&row)[NU]) {
  constexpr int V = 16 / sizeof(T);                     // elements per 16-byte read
  if constexpr ((NU % V) == 0) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const vec_t* p = reinterpret_cast<const vec_t*>(__builtin_assume_aligned(Lm + n * NU, 16));
#pragma unroll
    for (int i = 0; i < NU / V; ++i) {
      if (i * V <= n) {
        const vec_t v = p[i];
#pragma unroll
        for (int e = 0; e < V; ++e) row[i * V + e] = v[e];
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < NU; ++m)
      if (m <= n) row[m] = Lm[n * NU + m];
  }
}

// SRC_ACTIONS: z already holds raw actions (KMPPI), no colouring and no "+U"
template <typename T, int NU, bool DIAG, bool SRC_ACTIONS>
__device__ __forceinline__ void make_action(const ActionConsts<T, NU>& c, const T* __restrict__ Ut,
                                            const T* __restrict__ sampler_row /* this t, or null */,
                                            const T (&z)[NU], int orow, T (&v)[NU], T (&e)[NU]) {
  if constexpr (SRC_ACTIONS) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = z[n];
  } else if constexpr (DIAG) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = Ut[n] + (z[n] * c.sd[n] + c.mu[n]);
  } else {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      T Lr[NU];
      chol_row<T, NU>(c.Lm, n, Lr);
      T s = z[0] * Lr[0];
#pragma unroll
      for (int m = 1; m < NU; ++m)
        if (m <= n) s += z[m] * Lr[m];        // L = chol(Sigma) is lower triangular (mppi.py:139)
      v[n] = Ut[n] + (s + c.mu[n]);
    }
  }
  if (orow != -2) {
    if (orow == -1) {
#pragma unroll
      for (int n = 0; n < NU; ++n) v[n] = T(0);
    } else {
#pragma unroll
      for (int n = 0; n < NU; ++n) v[n] = sampler_row[n];
    }
  }
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    v[n] = clampT(v[n], c.lo[n], c.hi[n]);
    e[n] = (v[n] - Ut[n]) * c.e_scale;       // e_scale = 1 (MPPI) | 1/dt (SMPPI, mppi.py:544)
  }
}

template <typename T, int NU, bool DIAG>
__device__ __forceinline__ T action_cost_dot(const ActionConsts<T, NU>& c, const T* __restrict__ Ut,
                                             const T (&e)[NU]) {
  T nn[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) nn[n] = c.lambda_ * (c.abs_cost ? m_abs(e[n]) : e[n]);
  T acc = T(0);
  if constexpr (DIAG) {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      const T p = Ut[n] * (nn[n] * c.ci[n]);
      acc = (n == 0) ? p : acc + p;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      T s = nn[0] * c.Sm[n];
#pragma unroll
      for (int m = 1; m < NU; ++m) s += nn[m] * c.Sm[m * NU + n];
      const T p = Ut[n] * s;
      acc = (n == 0) ? p : acc + p;
    }
  }
  return acc;
}

// runtime-flag front ends for the kernels that are not specialised on DIAG / source
template <typename T, int NU>
__device__ __forceinline__ void make_action_rt(const KArgs<T>& a, const ActionConsts<T, NU>& c,
                                               const T* __restrict__ Ue, int t, const T (&z)[NU],
                                               int orow, T (&v)[NU], T (&e)[NU]) {
  const T* Ut = Ue + t * NU;
  const T* srow = orow >= 0 ? a.sampler + ((long long)orow * a.Tn + t) * NU : nullptr;
  if (a.noise_src == MPPI_NOISE_ACTIONS) make_action<T, NU, true, true>(c, Ut, srow, z, orow, v, e);
  else if (a.diag || a.coloured) make_action<T, NU, true, false>(c, Ut, srow, z, orow, v, e);
  else make_action<T, NU, false, false>(c, Ut, srow, z, orow, v, e);
}
// `Un` = the true nominal sequence (shift applied): U * action_cost uses U even when the noise
// is measured from another base (SMPPI)
template <typename T, int NU>
__device__ __forceinline__ T action_cost_dot_rt(const KArgs<T>& a, const ActionConsts<T, NU>& c,
                                                const T* __restrict__ Un, int t, const T (&e)[NU]) {
  return a.diag ? action_cost_dot<T, NU, true>(c, Un + t * NU, e)
                : action_cost_dot<T, NU, false>(c, Un + t * NU, e);
}

// compile-time layout of the TNK4 stream for a given NU: a super-step of TT timesteps consumes
// exactly P4 rows-of-4
template <int NU>
struct Stream {
  static constexpr int G = (NU % 4 == 0) ? 4 : ((NU % 2 == 0) ? 2 : 1);   // gcd(4, NU)
  static constexpr int P4 = NU / G;   // rows-of-4 per super-step
  static constexpr int TT = 4 / G;    // timesteps per super-step
};

}  // namespace mppi
