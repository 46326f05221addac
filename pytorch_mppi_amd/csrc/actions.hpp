// actions.hpp -- per-sample, per-timestep action pipeline shared by K1 (fused rollout),
// mppi_prepare (generic path / lazy attributes) and K3 (weighted update):
//   eps  = z*sqrt(diag)+mu | L z + mu                         mppi.py:201-206
//   v    = U[t] + eps                                          :380
//   v    = 0 (global row 0, sample_null_action) | sampler row  :387-400
//   v    = clamp(v, u_min, u_max)                              :383, :419-420
//   e    = v - U[t]              (post-clamp noise)            :385
//   a    = lambda*e*diag^-1 | (lambda*e) Sigma^-1 (|e| if abs) :186-199
//   pert += sum_n U[t,n]*a[n]                                  :415
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

// which overwrite row (if any) global sample kg is: returns -2 none, -1 null action, >=0 sampler row
template <typename T>
__device__ __forceinline__ int overwrite_row(const KArgs<T>& a, long long kg) {
  if (a.null_action && kg == 0) return -1;
  const long long r = kg - (a.null_action ? 1 : 0);
  if (r >= 0 && r < a.n_sampler) return (int)r;
  return -2;
}

template <typename T, int NU>
__device__ __forceinline__ void make_action(const KArgs<T>& a, const T* __restrict__ Ue /* [J] */,
                                            int t, const T (&z)[NU], int orow, T (&v)[NU],
                                            T (&e)[NU]) {
  const T* __restrict__ Ut = Ue + t * NU;
  if (a.noise_src == MPPI_NOISE_ACTIONS) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = z[n];
  } else if (a.diag) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = Ut[n] + (z[n] * a.L[n * NU + n] + a.mu[n]);
  } else {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      T s = z[0] * a.L[n * NU];
#pragma unroll
      for (int m = 1; m < NU; ++m) s += z[m] * a.L[n * NU + m];
      v[n] = Ut[n] + (s + a.mu[n]);
    }
  }
  if (orow == -1) {
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = T(0);
  } else if (orow >= 0) {
    const T* __restrict__ sa = a.sampler + ((long long)orow * a.Tn + t) * NU;
#pragma unroll
    for (int n = 0; n < NU; ++n) v[n] = sa[n];
  }
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    v[n] = clampT<T>(v[n], a.umin[n], a.umax[n]);
    e[n] = v[n] - Ut[n];
  }
}

template <typename T, int NU>
__device__ __forceinline__ T action_cost_dot(const KArgs<T>& a, const T* __restrict__ Ue, int t,
                                             const T (&e)[NU]) {
  const T* __restrict__ Ut = Ue + t * NU;
  T nn[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) nn[n] = a.lambda_ * (a.abs_cost ? m_abs(e[n]) : e[n]);
  T acc = T(0);
  if (a.diag) {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      const T p = Ut[n] * (nn[n] * a.sinv[n * NU + n]);
      acc = (n == 0) ? p : acc + p;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      T s = nn[0] * a.sinv[n];
#pragma unroll
      for (int m = 1; m < NU; ++m) s += nn[m] * a.sinv[m * NU + n];
      const T p = Ut[n] * s;
      acc = (n == 0) ? p : acc + p;
    }
  }
  return acc;
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
