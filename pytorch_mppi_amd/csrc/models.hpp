// models.hpp -- native dynamics/cost models as device functors.
//
// The reference's plugin API is "any Python callable" dynamics(state,u[,t]) / running_cost(state,u[,t])
// / terminal_state_cost(states,actions) (mppi.py:63-64, :314, :318, :325).  A callable cannot be
// fused into a kernel, so each native model is the same formula written once more as a device
// functor; pytorch_mppi_amd/models.py holds the matching torch callable + parameter blob.
// Contract (all per sample, state in registers):
//   step(x, u, t)   x <- f(x, u, t)           u is already u_scale * clamp(U + eps)  (mppi.py:313)
//   cost(x, u, t)   running cost on the POST-dynamics state (mppi.py:314,318)
//   terminal(x)     terminal cost of the last state (mppi.py:324-328; every terminal cost in the
//                   reference tree reads only states[..., -1, :])
#pragma once
#include "common.hpp"

namespace mppi {

// --- Pendulum: reference tests/pendulum.py:30-60 (gym Pendulum-v1 true dynamics) -------------
template <typename T>
struct PendulumModel {
  static constexpr int NX = 2, NU = 1;
  __device__ explicit PendulumModel(const KArgs<T>&) {}
  __device__ __forceinline__ void step(T (&x)[NX], const T (&u)[NU], int) const {
    const T uc = clampT(u[0], T(-2), T(2));                         // pendulum.py:41-42
    T nthd = x[1] + (T(15) * m_sin_moderate(x[0]) + T(3) * uc) * T(0.05);   // :44  3g/(2l)=15, 3/(ml^2)=3
    nthd = clampT(nthd, T(-8), T(8));                               // :45
    x[0] = x[0] + nthd * T(0.05);                                      // :46
    x[1] = nthd;
  }
  __device__ __forceinline__ T cost(const T (&x)[NX], const T (&)[NU], int) const {
    // angle_normalize (pendulum.py:52-53) with torch's floor-mod `%`
    const T pi = T(3.141592653589793), two_pi = T(6.283185307179586);
    const T an = m_floormod(x[0] + pi, two_pi) - pi;
    return an * an + T(0.1) * (x[1] * x[1]);                           // :56-61
  }
  __device__ __forceinline__ T terminal(const T (&)[NX]) const { return T(0); }
};

// --- n-D integrator ("quad-toy"): reference tests/benchmark_mppi.py:65-78 ---------------------
template <typename T, int NX_, int NU_>
struct IntegratorModel {
  static constexpr int NX = NX_, NU = NU_;
  __device__ explicit IntegratorModel(const KArgs<T>&) {}
  __device__ __forceinline__ void step(T (&x)[NX], const T (&u)[NU], int) const {
#pragma unroll
    for (int n = 0; n < NU && n < NX; ++n) x[n] = x[n] + u[n];         // :67-70
  }
  __device__ __forceinline__ T cost(const T (&x)[NX], const T (&)[NU], int) const {
    // (explicit fused multiply-adds: under -ffp-contract=fast the compiler decides per CONTEXT whether `c += x * x` becomes one fma
    //  or a packed multiply and an add -- the one-wave and the two-wave on-chip kernels came out differently, one or two ulps of
    //  cost_total apart, round 6.  Spelled out, every kernel that rolls this model out forms the same bits)
    T c = x[0] * x[0];
#pragma unroll
    for (int i = 1; i < NX; ++i) c = m_fma(x[i], x[i], c);             // :74-76
    return c;
  }
  __device__ __forceinline__ T terminal(const T (&)[NX]) const { return T(0); }
};

// --- linear dynamics + quadratic goal cost: reference tests/test_mppi.py:25-51 ----------------
// blob: B (NX,NU) row-major | goal (NX)
template <typename T, int NX_, int NU_>
struct LinearGoalModel {
  static constexpr int NX = NX_, NU = NU_;
  const T* __restrict__ B;
  const T* __restrict__ goal;
  __device__ explicit LinearGoalModel(const KArgs<T>& a) : B(a.mp), goal(a.mp + NX * NU) {}
  __device__ __forceinline__ void step(T (&x)[NX], const T (&u)[NU], int) const {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T d = u[0] * B[i * NU];
#pragma unroll
      for (int n = 1; n < NU; ++n) d += u[n] * B[i * NU + n];          // action @ B.T, :28-29
      x[i] = x[i] + d;
    }
  }
  __device__ __forceinline__ T goal_dist(const T (&x)[NX]) const {
    T c = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const T dx = goal[i] - x[i];
      c = (i == 0) ? dx * dx : c + dx * dx;                            // :40-42
    }
    return c;
  }
  __device__ __forceinline__ T cost(const T (&x)[NX], const T (&)[NU], int) const { return goal_dist(x); }
  __device__ __forceinline__ T terminal(const T (&x)[NX]) const { return goal_dist(x); }   // :49-51
};

// --- 2-layer MLP residual dynamics, per-lane VALU form (any hidden width) ---------------------
// x' = x + s * (W2 tanh(W1 [x;u] + b1) + b2), cost = sum_i qx_i x_i^2 + sum_n qu_n u_n^2  (qx = 1, qu = 0: the plain sum x^2 of
// BASELINE.json configs[3..4]; shape after reference tests/pendulum_approximate.py:47-67; the diagonal quadratic cost since round 5)
// blob: W1 (H, NX+NU) row-major | b1 (H) | W2 (NX, H) row-major | b2 (NX) | s (1) | qx (NX) | qu (NU)
// Weights are wave-uniform -> scalar loads; the MFMA formulation lives in rollout_mlp_mfma.hip.
template <typename T, int NX_, int NU_>
struct MlpModel {
  static constexpr int NX = NX_, NU = NU_, NI = NX_ + NU_;
  static constexpr bool NO_KMPPI_FUSE = true;   // the per-lane MLP needs its registers for the hidden layer
  const T* __restrict__ W1;
  const T* __restrict__ b1;
  const T* __restrict__ W2;
  const T* __restrict__ b2;
  const T* __restrict__ qx;
  const T* __restrict__ qu;
  T s;
  int H;
  __device__ explicit MlpModel(const KArgs<T>& a) {
    H = a.hidden;
    W1 = a.mp;
    b1 = W1 + (long long)H * NI;
    W2 = b1 + H;
    b2 = W2 + (long long)NX * H;
    s = b2[NX];
    qx = b2 + NX + 1;
    qu = qx + NX;
  }
  __device__ __forceinline__ void step(T (&x)[NX], const T (&u)[NU], int) const {
    T o[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) o[i] = T(0);
    for (int h = 0; h < H; ++h) {
      const T* __restrict__ w = W1 + (long long)h * NI;
      T acc = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) acc = m_fma(x[i], w[i], acc);
#pragma unroll
      for (int n = 0; n < NU; ++n) acc = m_fma(u[n], w[NX + n], acc);
      const T th = m_tanh(acc + b1[h]);
#pragma unroll
      for (int i = 0; i < NX; ++i) o[i] = m_fma(th, W2[(long long)i * H + h], o[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x[i] + s * (o[i] + b2[i]);
  }
  __device__ __forceinline__ T cost(const T (&x)[NX], const T (&u)[NU], int) const {
    // (qx = 1, qu = 0 give the bits of the plain sum of squares: 1 * x is x, and + 0 changes nothing)
    T c = (qx[0] * x[0]) * x[0];
#pragma unroll
    for (int i = 1; i < NX; ++i) c += (qx[i] * x[i]) * x[i];
#pragma unroll
    for (int n = 0; n < NU; ++n) c += (qu[n] * u[n]) * u[n];
    return c;
  }
  __device__ __forceinline__ T terminal(const T (&)[NX]) const { return T(0); }
};

}  // namespace mppi
