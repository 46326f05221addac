// common.hpp -- kernel argument block, Philox4x32-10 + Box-Muller, wave64 reductions.
// gfx950 only: wavefront = 64 lanes is hard-coded throughout.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mppi_amd.h"

namespace mppi {

// internal status of a K1 launcher: the launch carried the whole command (K3 + K4 included)
constexpr int MPPI_OK_FUSED = -100;
// ... or K1 + the workgroups' partial records (rollout_onchip.hpp): finalize_blocks_kernel completes the command
constexpr int MPPI_OK_ONCHIP = -101;
constexpr int MPPI_OK_KMPPI_W = -102;  // rollout_kmppi_kernel left partial records of the theta update: finalize_blocks on the theta problem

constexpr int WAVE = 64;
constexpr int BLOCK = 256;            // threads per workgroup of the per-sample kernels
constexpr int UPD_TJ = 64;            // j-columns per K3 tile (= one value per lane after the
                                      // transposing wave reduction)

// typed, by-value kernel argument block (lives in kernarg memory -> SGPRs)
template <typename T>
struct KArgs {
  int K, Tn, nx, nu, J, J4;
  long long k_offset;
  long long zp;   // row pitch of the TNK4 noise array in samples (>= K; mppi_noise_pitch)
  int model_id, diag, abs_cost, null_action, n_sampler, state_per_sample, shift, use_terminal,
      noise_src, u_per_command, hidden,
      coloured;   // z holds eps = L z + mu already (generator-side colouring): add U, bound, done
  T lambda_, u_scale, e_scale, smooth_w;
  unsigned long long seed, call;
  const T *state, *U, *u_init, *mu, *L, *sinv, *umin, *umax, *mp, *z, *sampler, *B;
  T *cost, *omega, *wnz, *U_out, *action_out, *pa, *noise, *pert, *states, *record;
  // workspace carve-up
  T* block_min;   // [nb1] minima of cost_total per 64 consecutive samples
  T* eta_part;    // [nkc]
  T* P_part;      // [nkc][Jpad]
  int nb1, nkc, Jpad, R;
  int n_env;      // MPPI_Batched: environments on grid.z (1 = single controller)
  unsigned long long* tstamp;   // measurement hook: {min entry, max exit} on wall_clock64, or null
  int M;                        // state rollouts per action sequence (mppi.py:334-373); 1 = the common case
  T var_cost, var_disc;         // M > 1: weight / per-step discount of the cost variance over the M rollouts
  const T* proc_sd;             // (nx) std of the native model's process noise, or null
  int fuse;                     // -1: K1 only | 0 / 1: whole command in K1's launch if eligible (value = K4's `apply`)
  unsigned* ticket;             // arrival counter of the single-launch command (workspace tail, kept at 0)
  const T *W, *theta;           // KMPPI inside K1 (rollout_kmppi.hpp): (T,S) operator, (S,nu) control points; else null
  int S;                        //   number of support points; z / seed / call then describe the SUPPORT-point stream
  int kw;                       //   1: the kernel also reduces its workgroups' part of the theta update (mppi.py:679): one partial
                                //   record {beta_b, eta_b, P_b[S*nu]} per 256 samples, on-chip carve with row stride kw_jpad
  int kw_jpad;
  int model_flags;              // MPPI_MODEL_FLAG_* (include/mppi_amd.h)
  T* spill;                     // on-chip command (rollout_onchip.hpp): where bounded noise that fits neither registers nor LDS waits for
  long long spill_cap;          //   its sample's weight, [row][padded sample][4], spill_cap elements; null / 0: generated a second time
  int seven;                    // the engine's generator runs Philox4x32-7 instead of -10 (MppiProblem.philox_rounds == 7; rng="philox7")
};

// Measurement hook (mppi_profile_enable, capi.hip): every workgroup stamps its entry and its exit on the device's wall clock
// into ONE OF STAMP_SLOTS {min entry, max exit} pairs of the launch (slot = workgroup index mod STAMP_SLOTS), the host takes
// the minimum / maximum over the slots.  One pair per launch -- what this was until round 4 -- put 2 x 256 atomics on one
// address: ~12 ns apiece at the memory side, 6-7 us added to every launch of an 80 us kernel (tools/edge_overhead.py:
// 87.8 us per C3 command with the stamps, 81.7 without).  With a slot per workgroup the atomics do not meet.
constexpr int STAMP_SLOTS = 256;
// slot = {max over the workgroups of ~entry, max of exit}: both stamps are atomicMax, so a window of slots is armed by zero-filling
// it (one memset; the {~0, 0} pattern of the min / max form needed an 8 MB host copy per window -- a millisecond in which the GPU
// idled and clocked down right in front of the region to be timed)
__device__ __forceinline__ void stamp_entry(unsigned long long* ts) {
  if (ts != nullptr && threadIdx.x == 0)
    atomicMax(&ts[2 * ((blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & (STAMP_SLOTS - 1))], ~(unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_exit(unsigned long long* ts) {      // call behind a block barrier: the last wave's exit
  if (ts != nullptr && threadIdx.x == 0)
    atomicMax(&ts[2 * ((blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & (STAMP_SLOTS - 1)) + 1], (unsigned long long)wall_clock64());
}

#ifndef MPPI_ONCHIP_PBROWS
#define MPPI_ONCHIP_PBROWS 12   // rows generated together by the on-chip command (csrc/rollout_onchip.hpp OnChip<NU>::PB)
#endif
#ifndef MPPI_ONCHIP_NTA
#define MPPI_ONCHIP_NTA 5   // weighting tiles the on-chip command keeps in registers (csrc/rollout_onchip.hpp OnChip<NU>::NTA)
#endif
// LDS carve of the kernel (floats): Ue[Jp] Um[Jp] G[Jp] | red[4] | ex[4][ntiles*64] | keepL[nsl*P4][256][4],
// Jp = the horizon padded to whole super-steps (a multiple of 4: every part starts on a 16-byte boundary)
struct OnChipLds {
  int Jp, ntiles, nsl, P4, nfac;     // nfac: 2 * nu * nu (chol(Sigma) | Sigma^-1) for a full Sigma, else 0
  __host__ __device__ int tables() const { return 3 * Jp + 4 + ((nfac + 3) & ~3); }
  __host__ __device__ int ex() const { return 4 * ntiles * 64; }
  __host__ __device__ size_t bytes() const { return ((size_t)tables() + ex()) * 4 + (size_t)nsl * P4 * 256 * 16; }
};

// Where the rows of a sample wait for its weight, as a function of (nu, horizon): the first AG_SS super-steps in registers, the next
// `nsl` (whole weighting tiles, as many as fit the 160 KB beside the tables) in LDS, the remaining `nsm` (whole tiles, the horizon's
// last partial one included) in the caller's spill array if there is one -- else they are generated a second time.  One function for
// the launcher and for mppi_onchip_spill_elems (the size a caller allocates); the constants are OnChip<NU>'s, restated at run time.
struct OnChipGeometry {
  int P4, TT, SW, AG_SS, nss, ntiles, nsl, nsm;
  size_t smem;
  bool ok;
};
static inline OnChipGeometry onchip_geometry(int nu, int Tn, bool diag) {
  OnChipGeometry g{};
  const int G = (nu % 4 == 0) ? 4 : ((nu % 2 == 0) ? 2 : 1);
  g.P4 = nu / G;
  g.TT = 4 / G;
  g.SW = (16 / g.P4) > 0 ? (16 / g.P4) : 1;
  g.AG_SS = MPPI_ONCHIP_NTA * g.SW;
  const int RG = g.P4 >= 3 ? 1 : (g.P4 == 2 ? 2 : 4);
  g.nss = (Tn + g.TT - 1) / g.TT;
  g.ntiles = (g.nss + g.SW - 1) / g.SW;
  OnChipLds L{g.nss * g.P4 * 4, g.ntiles, 0, g.P4, diag ? 0 : 2 * nu * nu};
  g.ok = g.P4 <= 16 && (g.SW % RG) == 0 && L.bytes() <= 160 * 1024;
  if (!g.ok) return g;
  const long long room = (160 * 1024 - (long long)L.bytes()) / ((long long)g.P4 * 256 * 16);
  int nsl = g.nss - g.AG_SS;
  if (nsl < 0) nsl = 0;
  if (nsl > room) nsl = (int)room;
  nsl -= nsl % g.SW;
  L.nsl = g.nsl = nsl;
  g.nsm = g.ntiles * g.SW - g.AG_SS - nsl;
  if (g.nsm < 0) g.nsm = 0;
  g.smem = L.bytes();
  return g;
}

// The same command with TWO waves per sample group (csrc/rollout_onchip_pair.hpp): the super-steps go to the two waves of a pair in
// alternating chunks of CH; each wave keeps ITS rows -- the first KR local super-steps in registers, the next `nsl` (whole tiles) in LDS,
// the remaining `nsm` in the spill array ([local row][workgroup thread of 512][4]).  `nls` counts the local super-steps of the wave
// that owns the even chunks (the other one has as many or one chunk fewer), `ntl` its weighting tiles.
#ifndef MPPI_ONCHIP_PAIR_DEFAULT
#define MPPI_ONCHIP_PAIR_DEFAULT 1   // MPPI_ONCHIP_PAIR unset: 0 = the one-wave kernel, 1 = the pair kernel wherever it applies
#endif
#ifndef MPPI_PAIR_KT
#define MPPI_PAIR_KT 1      // weighting tiles per wave kept in registers (each wave has 256 registers, not 512: two tiles spill)
#endif
struct OnChipPairGeometry {
  int P4, TT, SW, CH, KR, nss, nch, nit, nls, ntl, nsl, nsm, shn;
  size_t smem;
  bool ok;
};
// plain: the command without |noise| cost, u_scale, SMPPI terms; otherwise the previous timestep's action rides in the hand-over too
static inline OnChipPairGeometry onchip_pair_geometry(int nu, int nx, int Tn, bool plain = true) {
  OnChipPairGeometry g{};
  const int G = (nu % 4 == 0) ? 4 : ((nu % 2 == 0) ? 2 : 1);
  g.P4 = nu / G;
  g.TT = 4 / G;
  g.SW = (16 / g.P4) > 0 ? (16 / g.P4) : 1;
  g.CH = (MPPI_ONCHIP_PBROWS / g.P4) > 0 ? (MPPI_ONCHIP_PBROWS / g.P4) : 1;
  if (!plain && g.CH >= 2) --g.CH;                                    // (OnChipPair<NU, false>::CH)
  g.KR = MPPI_PAIR_KT * g.SW;
  g.nss = (Tn + g.TT - 1) / g.TT;
  g.nch = (g.nss + g.CH - 1) / g.CH;
  g.nit = (g.nch + 1) / 2;
  g.nls = g.nit * g.CH;
  g.ntl = (g.nls + g.SW - 1) / g.SW;
  const int hand = (nx + 2 + (plain ? 0 : nu)) * BLOCK, ex = 8 * g.ntl * 64;
  g.shn = hand > ex ? hand : ex;                                      // floats: the state hand-over, later the waves' column sums
  const long long fixed = ((long long)3 * g.nss * g.P4 * 4 + 16 + g.shn) * 4;
  g.ok = g.P4 <= 16 && fixed <= 160 * 1024;
  if (!g.ok) return g;
  const long long room = (160 * 1024 - fixed) / ((long long)g.P4 * 2 * BLOCK * 16);
  int nsl = (g.nch / 2) * g.CH - g.KR;                                // only what BOTH waves of a pair really write may wait in LDS
  if (nsl < 0) nsl = 0;
  if (nsl > room) nsl = (int)room;
  nsl -= nsl % g.SW;
  g.nsl = nsl;
  // The wave's last `dls` local super-steps are generated a second time instead of waiting in memory: the weighting phase is bound
  // either by the fetch of the memory rows (out of the Infinity Cache, chip-wide) or by its VALU work (column sums + second generation,
  // two waves per SIMD), and the two proceed side by side.  Measured at C3 (profiles/r06_k_onchip_pair_check.txt; 22 real super-steps
  // beyond registers and LDS, 3 of padding): dls = 0 / 5 / 10 / 15 -> 67.1 / 62.0 / 60.4 / 62.9 us, splits inside a tile (6, 7: 62.3,
  // 62.4) no better than their neighbours -- so whole tiles, as many as bring the regenerated share closest to a third of the real
  // super-steps out there (C3: the last two tiles = 7 of 22).  MPPI_PAIR_DLS in the environment overrides it (measurements; any value).
  const int after = g.ntl * g.SW - g.KR - nsl;                        // local super-steps beyond registers and LDS, padding included
  static const int forced = [] { const char* e = getenv("MPPI_PAIR_DLS"); return (e && *e) ? atoi(e) : -1; }();
  int dls = 0;
  if (after > 0) {
    auto real = [&](int ls) { return ls < g.nls && ((2 * (ls / g.CH)) * g.CH + ls % g.CH) < g.nss; };
    int areal = 0;
    for (int ls = g.KR + nsl; ls < g.ntl * g.SW; ++ls) areal += real(ls) ? 1 : 0;
    double best = 1e30;
    for (int c = 0; c <= 3 && c * g.SW <= after; ++c) {
      int dreal = 0;
      for (int ls = (g.ntl - c) * g.SW; ls < g.ntl * g.SW; ++ls) dreal += real(ls) ? 1 : 0;
      const double d = dreal - 0.32 * areal;
      if ((d < 0 ? -d : d) < best - 1e-9) { best = d < 0 ? -d : d; dls = c * g.SW; }
    }
    if (forced >= 0) dls = forced < after ? forced : after;
  }
  g.nsm = (after > 0 ? after : 0) - dls;
  g.smem = (size_t)fixed + (size_t)nsl * g.P4 * 2 * BLOCK * 16;
  return g;
}

// the workspace of an on-chip command: one partial record per 256-sample workgroup
//   block_min[0 .. nchunks)  beta_b | eta_part[0 .. nchunks)  eta_b | P_part[nchunks][Jpad]
// (capi.hip's carve() sizes the buffer for K/256 partial records whatever the chunking)
template <typename T>
__host__ __device__ inline void onchip_carve(KArgs<T>& a) {
  const int nchunks = (a.K + BLOCK - 1) / BLOCK;
  a.nkc = nchunks;
  a.R = 1;
  a.P_part = a.eta_part + nchunks;
}

// MPPI_Batched: the view of the argument block for environment blockIdx.z.  The noise (z), all
// parameters and the model are shared; state, nominal sequence, costs, weights, workspace and
// outputs are per environment.
template <typename T>
__device__ __forceinline__ KArgs<T> env_view(const KArgs<T>& a) {
  if (a.n_env <= 1) return a;
  const long long e = blockIdx.z;
  KArgs<T> b = a;
  b.state = a.state + e * a.nx;
  b.U = a.U + e * a.J;
  b.cost = a.cost + e * a.K;
  if (a.omega) b.omega = a.omega + e * a.K;
  if (a.wnz) b.wnz = a.wnz + e * a.K;
  if (a.U_out) b.U_out = a.U_out + e * a.J;
  if (a.action_out) b.action_out = a.action_out + e * a.u_per_command * a.nu;
  if (a.pa) b.pa = a.pa + e * a.K * a.J;
  if (a.noise) b.noise = a.noise + e * a.K * a.J;
  if (a.pert) b.pert = a.pert + e * a.K;
  if (a.record) b.record = a.record + e * (2 + a.J);
  b.block_min = a.block_min + e * a.nb1;
  b.eta_part = a.eta_part + e * a.nkc;
  b.P_part = a.P_part + e * (long long)a.nkc * a.Jpad;
  return b;
}

// ---------------------------------------------------------------------------------------------
// scalar math overloads (accurate ocml forms: parity against the CPU oracle is the first gate)
// ---------------------------------------------------------------------------------------------
// sinf / cosf without the library routine's branches: a rollout is ONE dependent chain per lane, and at one wave per SIMD every
// instruction of the step's chain is paid in latency -- ocml's sinf is ~45 of them behind two exec-mask branches (the pendulum's
// step: 8 of the single-launch command's 16 us at C2).  Cody-Waite reduction by pi/2 in three parts (exact products with k for
// |x| < 1e5) and the usual minimax polynomials on [-pi/4, pi/4]: 22 instructions, <= 1.52 ulp against the correctly rounded
// result on 1.6e7 random arguments in |x| < 1e5 (numpy's own float32 sin: 1.47) -- what the reference's torch.sin gives or takes.
// Larger arguments are first reduced modulo 2 pi in fp64 (a handful of instructions; fp32-level accuracy up to |x| ~ 2^40) -- NOT
// handed to the library routine: its Payne-Hanek path brings 400 B of scratch per lane into every kernel that could reach it.
// NaN for NaN / inf.
__device__ __forceinline__ void sincos_reduce(float x, float& s, float& c, int& q) {
  const float k = __builtin_rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.57079601287841796875f, x);
  r = fmaf(-k, 3.1391647326017846e-07f, r);
  r = fmaf(-k, 5.3903029534742384e-15f, r);
  q = (int)k;
  const float r2 = r * r;
  float ps = fmaf(-1.95152959e-4f, r2, 8.33216087e-3f);
  ps = fmaf(ps, r2, -1.66666546e-1f);
  s = fmaf(ps * r2, r, r);
  float pc = fmaf(2.44331571e-5f, r2, -1.38873163e-3f);
  pc = fmaf(pc, r2, 4.16666457e-2f);
  pc = fmaf(pc, r2, -0.5f);
  c = fmaf(pc, r2, 1.0f);
}
__device__ __forceinline__ float sincos_moderate(float x) {
  if (__builtin_expect(!(__builtin_fabsf(x) <= 1.0e5f), 0)) {
    const double t = (double)x;
    const double k = __builtin_rint(t * 0.15915494309189535);            // 1 / (2 pi)
    // 2 pi in two parts (hi = the nearest double, lo = what it misses): with the single constant the reduced argument is off by
    // k * 2.4e-16 -- 4e-7 at 1e10, more than fp32's own 6e-8
    x = (float)__builtin_fma(-k, 2.4492935982947064e-16, __builtin_fma(-k, 6.283185307179586, t));   // |x| <= pi  (NaN for NaN / inf)
  }
  return x;
}
__device__ __forceinline__ float m_sin(float x) {
  x = sincos_moderate(x);
  float s, c;
  int q;
  sincos_reduce(x, s, c, q);
  const float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_cos(float x) {
  x = sincos_moderate(x);
  float s, c;
  int q;
  sincos_reduce(x, s, c, q);
  const float v = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -v : v;
}
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
// tanhf without branches (the ocml routine is ~30 instructions behind two exec-mask branches; a traced network calls it
// once per hidden unit and timestep).  |x| >= 0.25: t = e^(-2|x|), (1 - t) / (1 + t) on v_exp_f32 + v_rcp_f32 -- no
// cancellation there (1 - t >= 0.39), ~3 ulp; |x| < 0.25: the odd Taylor polynomial through x^9 (next term < 1e-8
// relative).  Both sides are evaluated, a select picks one; sign by v_bfi.
__device__ __forceinline__ float m_tanh(float x) {
  const float a = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_exp2f(a * -2.8853900817779268f);         // e^(-2a) = 2^(-2a log2 e)
  const float big = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  const float x2 = a * a;
  float p = fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f);
  p = fmaf(x2, p, 2.0f / 15.0f);
  p = fmaf(x2, p, -1.0f / 3.0f);
  const float small = fmaf(a * x2, p, a);
  return __builtin_copysignf(a < 0.25f ? small : big, x);
}
__device__ __forceinline__ double m_tanh(double x) { return tanh(x); }
// (the rest of the scalar vocabulary of traced callables, pytorch_mppi_amd/trace.py)
#define MPPI_M1(NAME, F32, F64)                                         \
  __device__ __forceinline__ float NAME(float x) { return F32(x); }      \
  __device__ __forceinline__ double NAME(double x) { return F64(x); }
MPPI_M1(m_erf, erff, erf) MPPI_M1(m_atan, atanf, atan) MPPI_M1(m_asin, asinf, asin) MPPI_M1(m_acos, acosf, acos)
MPPI_M1(m_sinh, sinhf, sinh) MPPI_M1(m_cosh, coshf, cosh) MPPI_M1(m_expm1, expm1f, expm1) MPPI_M1(m_log1p, log1pf, log1p)
MPPI_M1(m_ceil, ceilf, ceil) MPPI_M1(m_rint, rintf, rint) MPPI_M1(m_trunc, truncf, trunc)
#undef MPPI_M1
__device__ __forceinline__ float m_fmod(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double m_fmod(double a, double b) { return fmod(a, b); }
// Python / torch floor-mod `a % b` for b > 0 and |a / b| < 2^22: k = floor(a / b), r = a - k b.  fmod's
// result is exactly representable, so the single rounding of the fma is exact once k is right; the
// two fix-ups repair a k that the rounded quotient put off by one.  Same bits as fmodf + sign
// fix-up (the ocml fmod is an exact-remainder LOOP, ~100 instructions per call).
__device__ __forceinline__ float m_floormod(float a, float b) {
  const float k = floorf(a * (1.0f / b));
  float r = fmaf(-k, b, a);
  r = r < 0.0f ? r + b : r;
  r = r >= b ? r - b : r;
  return r;
}
__device__ __forceinline__ double m_floormod(double a, double b) {
  const double k = floor(a * (1.0 / b));
  double r = fma(-k, b, a);
  r = r < 0.0 ? r + b : r;
  r = r >= b ? r - b : r;
  return r;
}
// sinf for arguments of moderate size: ocml's sinf, with its Payne-Hanek path for huge arguments (|x| >= 2^17:
// 64-bit multiply chains, ~1000 instructions, the registers to match) pruned from the hot path by telling the
// compiler the range.  An argument that really is that large (no physical angle is) is first reduced modulo 2 pi in
// fp64 -- a handful of instructions, no call, no stack -- which keeps fp32-level accuracy up to |x| ~ 2^40; same bits
// as sinf below 8192, NaN for NaN / inf like sinf.
__device__ __forceinline__ float m_sin_moderate(float x) {
  float xr = x;
  if (__builtin_expect(!(__builtin_fabsf(x) < 8192.0f), 0)) {
    const double t = (double)x;
    const double k = __builtin_rint(t * 0.15915494309189535);            // 1 / (2 pi)
    xr = (float)__builtin_fma(-k, 6.283185307179586, t);                 // |xr| <= pi  (NaN for NaN / inf)
  }
  const float ax = __builtin_fabsf(xr);
  if (__builtin_expect(!(ax < 8192.0f), 0)) return xr - xr;              // NaN
  __builtin_assume(ax < 8192.0f);
  return sinf(xr);
}
__device__ __forceinline__ double m_sin_moderate(double x) { return sin(x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
// (used by the functor bodies pytorch_mppi_amd/trace.py writes from torch callables)
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_floor(float x) { return floorf(x); }
__device__ __forceinline__ double m_floor(double x) { return floor(x); }
__device__ __forceinline__ float m_pow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double m_pow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float m_atan2(float a, float b) { return atan2f(a, b); }
__device__ __forceinline__ double m_atan2(double a, double b) { return atan2(a, b); }
template <typename T> __device__ __forceinline__ T m_min(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ __forceinline__ T m_max(T a, T b) { return a > b ? a : b; }
__device__ __forceinline__ float m_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double m_fma(double a, double b, double c) { return fma(a, b, c); }

// torch.clamp(x, lo, hi) = min(max(x, lo), hi) for lo <= hi: the median of the three, ONE v_med3_f32 (fminf(fmaxf())
// is three instructions: the compiler canonicalises x first).  Infinite bounds pass x through; a NaN x comes out
// as lo in both forms (fmaxf and v_med3 both return the non-NaN operand).
__device__ __forceinline__ float clampT(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ double clampT(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

template <typename T> __device__ __forceinline__ T inf_v();
template <> __device__ __forceinline__ float inf_v<float>() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_v<double>() { return __builtin_huge_val(); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011; constants as in rocrand_philox4x32_10.h:62-65).
// counter = (k_global, jb, call_lo, call_hi), key = (seed_lo, seed_hi)  -- engine-defined.
// ---------------------------------------------------------------------------------------------
struct U4 { unsigned x, y, z, w; };

// a ^ b ^ c in ONE instruction: gfx950's v_bitop3_b32 (any 3-input bitwise function; 0x96 = the truth table of the 3-way
// XOR).  hipcc emits two v_xor_b32 for the C expression -- 40 of the ~100 VALU instructions of a Philox4x32-10 call, and
// the generator is what the on-chip command's time is made of.
__host__ __device__ __forceinline__ unsigned xor3(unsigned a, unsigned b, unsigned c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

#ifndef MPPI_PHILOX_ROUNDS
#define MPPI_PHILOX_ROUNDS 10     // measurement seam (tools/micro/onchip_parts.hip -DMPPI_PHILOX_ROUNDS=7); the product is Philox4x32-10
#endif
__host__ __device__ __forceinline__ void philox_round(U4& c, unsigned& k0, unsigned& k1, const bool first) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  unsigned long long p0 = (unsigned long long)M0 * c.x;
  unsigned long long p1 = (unsigned long long)M1 * c.z;
  U4 n;
  // round 0: c.z (the command number), c.y (the row) and the key are wave-uniform in every kernel whose lane is a sample,
  // so this word is scalar work (s_mul_hi / s_xor) when spelled with ^ -- v_bitop3_b32 has no scalar form: as a builtin it
  // put the word into a VGPR, and the KMPPI-fused K1, which hoists these row constants out of its chunk loop, spilled
  // 131 of them to scratch (94 us instead of 70.7)
  n.x = first ? ((unsigned)(p1 >> 32) ^ c.y ^ k0) : xor3((unsigned)(p1 >> 32), c.y, k0);
  n.y = (unsigned)p1;
  n.z = xor3((unsigned)(p0 >> 32), c.w, k1);
  n.w = (unsigned)p0;
  c = n;
  k0 += W0;
  k1 += W1;
}
template <int ROUNDS>
__host__ __device__ __forceinline__ U4 philox4x32_rounds(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) philox_round(c, k0, k1, r == 0);
  return c;
}
__host__ __device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) { return philox4x32_rounds<MPPI_PHILOX_ROUNDS>(c, k0, k1); }
// Philox4x32-R for R = 7 (`seven`, wave-uniform: rng="philox7") or 10, chosen at RUN time.  Seven rounds is the fewest that pass
// BigCrush (Salmon et al. 2011, table 2; Random123 ships it as philox4x32_R<7> and pins it with its own known-answer vectors:
// oracle/philox.py), ten is the conservative default of every library; the three rounds are 30 % of the generator's multiplies, which
// are what the on-chip command's time is made of (profiles/r05_philox_rounds.txt: K1 71.4 -> 66.4 us at C3) -- THAT kernel takes the
// round count as a template parameter.  Here: seven unrolled rounds, then a ROLLED loop of zero or three more behind a scalar
// counter -- one round of extra code and no extra live values.  (Two whole chains behind a branch, and also a shared prefix with
// three unrolled rounds behind a branch, cost kernels that generate in the lane their registers: K1 generating its own rows
// 0.102 -> 0.129 ms at C3, 241 spilled registers in the widest KMPPI interpolation kernel.)  Kernels whose allocation does not
// survive even this keep ten rounds at compile time (noise4<.., 10>) and their launchers refuse seven-round problems.
__host__ __device__ __forceinline__ U4 philox4x32_r(U4 c, unsigned k0, unsigned k1, const bool seven) {
  philox_round(c, k0, k1, true);
#pragma unroll
  for (int r = 1; r < 7; ++r) philox_round(c, k0, k1, false);
  const int extra = seven ? 0 : MPPI_PHILOX_ROUNDS - 7;
#pragma unroll 1
  for (int r = 0; r < extra; ++r) philox_round(c, k0, k1, false);
  return c;
}

// Box-Muller on two 32-bit words -> two N(0,1) floats.  u1 in (0,1], u2 in [0,1).
// v_sin/v_cos take their argument in revolutions, so 2*pi*u2 is never formed.
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
  float u1 = fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);  // 2^-32, 2^-33
  float u2 = (float)b * 2.3283064365386963e-10f;
  // r = sqrt(-2 ln u1) = sqrt(-2 ln2 * log2 u1): raw v_log_f32 / v_sqrt_f32 (1 ulp each; the
  // argument is in [0, 44.4], never denormal, so the slow-path fix-ups of sqrtf/logf are dead weight)
  float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// ROUNDS: 0 = the round count is `seven`'s, at run time | 7 / 10 = fixed at compile time
template <typename T, int ROUNDS = 0>
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned long long call,
                                               long long kg, long long jb, T (&out)[4], const bool seven = false) {
  U4 c{(unsigned)kg, (unsigned)jb, (unsigned)call, (unsigned)(call >> 32) ^ (unsigned)(kg >> 32)};
  U4 r;
  if constexpr (ROUNDS == 10) r = philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  else if constexpr (ROUNDS == 7) r = philox4x32_rounds<7>(c, (unsigned)seed, (unsigned)(seed >> 32));
  else r = philox4x32_r(c, (unsigned)seed, (unsigned)(seed >> 32), seven);
  float a, b, d, e;
  box_muller(r.x, r.y, a, b);
  box_muller(r.z, r.w, d, e);
  out[0] = (T)a; out[1] = (T)b; out[2] = (T)d; out[3] = (T)e;
}

// Process noise of the fused multi-rollout K1 (rollout.hpp, rollout_stream_multi): its own Philox key (seed ^ tag)
// and counter (sample, (t * PROCESS_NOISE_MM + m) * ceil(nx/4) + block, command)
constexpr unsigned long long PROCESS_NOISE_KEY_TAG = 0x5A5A5A5AA5A5A5A5ull;
constexpr int PROCESS_NOISE_MM = 4;

// one row-of-4 of the noise stream for (jb, local sample k)
template <typename T>
__device__ __forceinline__ void load4(const T* __restrict__ z, long long K, long long jb, int k,
                                      T (&out)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* __restrict__ z, long long K,
                                             long long jb, int k, float (&out)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(z + (jb * K + k) * 4);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load4<double>(const double* __restrict__ z, long long K,
                                              long long jb, int k, double (&out)[4]) {
  const double2* p = reinterpret_cast<const double2*>(z + (jb * K + k) * 4);
  const double2 a = p[0], b = p[1];
  out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y;
}

// the LAST read of a row (K3): non-temporal, so that the rows do not stay in L2 / Infinity Cache
// behind their final use -- the next command's freshly written draw keeps that room instead
// (measured at C3: K3 34.5 -> 34.0 us and the following K1 35.1 -> 33.1 us)
template <typename T>
__device__ __forceinline__ void load4_last(const T* __restrict__ z, long long K, long long jb, int k, T (&out)[4]) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const vec_t* p = reinterpret_cast<const vec_t*>(z + (jb * K + k) * 4);
#pragma unroll
  for (int i = 0; i < 4 / V; ++i) {
    const vec_t v = __builtin_nontemporal_load(p + i);
#pragma unroll
    for (int e = 0; e < V; ++e) out[i * V + e] = v[e];
  }
}

// store one generated row-of-4 into the TNK4 array (Philox "generate once, re-read in K3" mode)
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ z, long long K, long long jb, int k, const T (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float* __restrict__ z, long long K, long long jb, int k,
                                              const float (&v)[4]) {
  *reinterpret_cast<float4*>(z + (jb * K + k) * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<double>(double* __restrict__ z, long long K, long long jb, int k,
                                               const double (&v)[4]) {
  double2* p = reinterpret_cast<double2*>(z + (jb * K + k) * 4);
  p[0] = make_double2(v[0], v[1]);
  p[1] = make_double2(v[2], v[3]);
}

// ROUNDS: 0 = the generator's round count is the launch's (a.seven, wave-uniform) | 10 = fixed at compile time (kernels whose
// register allocation does not survive the run-time choice: their launcher refuses seven-round problems)
template <typename T, int NOISE, int ROUNDS = 0>
__device__ __forceinline__ void noise4(const KArgs<T>& a, long long jb, int k, T (&out)[4]) {
  static_assert(NOISE != MPPI_NOISE_KTN, "the (K,T,nu) layout has its own loaders");
  if constexpr (NOISE == MPPI_NOISE_PHILOX) {
    philox_normal4<T, ROUNDS>(a.seed, a.call, a.k_offset + k, jb, out, a.seven != 0);
  } else {
    load4<T>(a.z, a.zp, jb, k, out);
  }
}
// K3's form: same rows, read for the last time
template <typename T, int NOISE>
__device__ __forceinline__ void noise4_last(const KArgs<T>& a, long long jb, int k, T (&out)[4]) {
  if constexpr (NOISE == MPPI_NOISE_PHILOX) {
    philox_normal4<T>(a.seed, a.call, a.k_offset + k, jb, out, a.seven != 0);
  } else {
    load4_last<T>(a.z, a.zp, jb, k, out);
  }
}

// ---------------------------------------------------------------------------------------------
// wave64 / block reductions (fixed order -> deterministic)
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    T o = __shfl_xor(v, m, WAVE);
    v = o < v ? o : v;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}

// block-wide min / sum over BLOCK threads; result valid in every thread.  `sm` >= BLOCK/WAVE
template <typename T>
__device__ __forceinline__ T block_min(T v, T* sm) {
  v = wave_min(v);
  const int w = threadIdx.x / WAVE;
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) sm[w] = v;
  __syncthreads();
  T r = sm[0];
#pragma unroll
  for (int i = 1; i < BLOCK / WAVE; ++i) r = sm[i] < r ? sm[i] : r;
  return r;
}
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x / WAVE;
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) sm[w] = v;
  __syncthreads();
  T r = sm[0];
#pragma unroll
  for (int i = 1; i < BLOCK / WAVE; ++i) r += sm[i];
  return r;
}

// Transposing wave reduction: every lane holds v[0..64); afterwards lane l holds
// sum over lanes of v[l].  63 shuffles per lane instead of 64*6.
template <typename T>
__device__ __forceinline__ T wave_reduce_transpose64(T (&v)[64]) {
  const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const bool upper = (lane & s) != 0;
#pragma unroll
    for (int i = 0; i < s; ++i) {
      T lo = v[i], hi = v[i + s];
      // keep both operands in VGPRs: without this the compiler folds the lane-dependent select
      // into a dynamically indexed load of v[], which forces the whole array into scratch
      asm volatile("" : "+v"(lo), "+v"(hi));
      const T send = upper ? lo : hi;
      const T keep = upper ? hi : lo;
      v[i] = keep + __shfl_xor(send, s, WAVE);
    }
  }
  return v[0];
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// fp32: the same transposing wave reduction without the LDS crossbar (a different -- equally fixed -- summation order; every
// caller uses this one function, so paths that must agree bit for bit still do).  s = 32 / 16: v_permlane32_swap / v_permlane16_swap exchange the halves of a
// register PAIR in one instruction, so the keep/send selects disappear; s = 8 .. 1: pair sums through DPP
// (row_ror:8 == lane^8, row_half_mirror pairs across bit 2, quad_perm for lane^2 / lane^1) and one select.
// 141 VALU instructions per 64 columns against 189 + 63 ds_bpermute.
template <>
__device__ __forceinline__ float wave_reduce_transpose64<float>(float (&v)[64]) {
  const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 32]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float X = v[i] + dpp_mov<0x128>(v[i]), Y = v[i + 8] + dpp_mov<0x128>(v[i + 8]);
      v[i] = up ? Y : X;
    }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float X = v[i] + dpp_mov<0x141>(v[i]), Y = v[i + 4] + dpp_mov<0x141>(v[i + 4]);
      v[i] = up ? Y : X;
    }
  }
  {
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float X = v[i] + dpp_mov<0x4E>(v[i]), Y = v[i + 2] + dpp_mov<0x4E>(v[i + 2]);
      v[i] = up ? Y : X;
    }
  }
  {
    const bool up = (lane & 1) != 0;
    const float X = v[0] + dpp_mov<0xB1>(v[0]), Y = v[1] + dpp_mov<0xB1>(v[1]);
    v[0] = up ? Y : X;
  }
  return v[0];
}


}  // namespace mppi
