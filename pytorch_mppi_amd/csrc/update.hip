// update.hip -- everything of one command() that is not the fused rollout:
//   noise_fill_philox / noise_from_ktn   noise stream producers (TNK4 layout)
//   kmppi_interp                         KMPPI support points -> raw trajectories
//   prepare                              generic path + lazily materialised public attributes
//   cost_block_min                       generic path: minima of a host-assembled cost_total
//   weights_partial (K3)                 beta, w = exp(-(c-beta)/lambda), per-block eta and P
//   finalize (K4) / combine (K5)         fixed-order reductions, U update, action, omega
// All reductions are fixed-order (no float atomics): the same inputs give the same bits on every
// launch and on every rank.
#include "actions.hpp"
#include "update.hpp"
#include "weights.hpp"

namespace mppi {

// =============================================================================================
// noise producers
// =============================================================================================
template <typename T>
__global__ void __launch_bounds__(BLOCK) noise_fill_philox_kernel(const KArgs<T> a, T* __restrict__ out) {
  const int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= a.K) return;
  for (int jb = blockIdx.y; jb < a.J4; jb += gridDim.y) {
    T r[4];
    philox_normal4<T>(a.seed, a.call, a.k_offset + k, jb, r, a.seven != 0);
    T* o = out + ((long long)jb * a.zp + k) * 4;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
  }
}

// the same stream, coloured by the generator: eps[t] = chol(Sigma) z[t] + mu (mppi.py:201-206) for
// whole timesteps -- one thread per (sample, super-step of TT timesteps = P4 rows-of-4).  The
// factors are uniform (scalar loads); at full occupancy the nu*(nu+1)/2 FMAs per timestep cost a few
// microseconds, against ~25 us each in K1 and K3 where one wave per SIMD does them behind LDS reads.
template <typename T, int NU>
__global__ void __launch_bounds__(BLOCK) noise_fill_philox_coloured_kernel(const KArgs<T> a, T* __restrict__ out) {
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  const int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= a.K) return;
  const int nss = (a.Tn + TT - 1) / TT;
  // the factors once, before any store: uniform -> scalar loads, live in SGPRs / uniform registers
  T Lr[NU * (NU + 1) / 2], mr[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    mr[n] = a.mu[n];
#pragma unroll
    for (int m = 0; m <= n; ++m) Lr[n * (n + 1) / 2 + m] = a.L[n * NU + m];
  }
  for (int ss = blockIdx.y; ss < nss; ss += gridDim.y) {
    T zc[P4 * 4], ec[P4 * 4];
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      T r[4];
      philox_normal4<T>(a.seed, a.call, a.k_offset + k, (long long)ss * P4 + i, r, a.seven != 0);
      zc[4 * i] = r[0]; zc[4 * i + 1] = r[1]; zc[4 * i + 2] = r[2]; zc[4 * i + 3] = r[3];
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        T s = zc[tt * NU] * Lr[n * (n + 1) / 2];
#pragma unroll
        for (int m = 1; m < NU; ++m)
          if (m <= n) s += zc[tt * NU + m] * Lr[n * (n + 1) / 2 + m];     // lower triangular
        ec[tt * NU + n] = s + mr[n];
      }
    }
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      const T r[4] = {ec[4 * i], ec[4 * i + 1], ec[4 * i + 2], ec[4 * i + 3]};
      store4<T>(out, a.zp, (long long)ss * P4 + i, k, r);
    }
  }
}

// (K, J) row-major -> [J4][K][4].  Tile: 64 samples x 64 columns through LDS so that both the
// read (along j: 16 lanes x 16 B = one 256-B run per sample) and the write (along k: 64 lanes x
// 16 B = 1 KiB) are coalesced 16-byte accesses.  VEC = false: J % 4 != 0 (rows not 16-B aligned).
template <typename T, bool VEC>
__global__ void __launch_bounds__(BLOCK) noise_from_ktn_kernel(const KArgs<T> a, const T* __restrict__ in,
                                                               T* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) T tile[64][68];   // 68: rows stay 16-B aligned, 4-way -> conflict-light
  const int k0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  if constexpr (VEC) {
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;   // 16 column-quads x 16 rows per pass
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = r0 + 16 * p, k = k0 + r, j = j0 + 4 * c4;
      T v[4] = {T(0), T(0), T(0), T(0)};
      if (k < a.K && j < a.J) load4<T>(in + (long long)k * a.J + j, 0, 0, 0, v);   // J % 4 == 0: whole quad valid
      tile[r][4 * c4 + 0] = v[0]; tile[r][4 * c4 + 1] = v[1]; tile[r][4 * c4 + 2] = v[2]; tile[r][4 * c4 + 3] = v[3];
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int r = ty; r < 64; r += 4) {
      const int k = k0 + r, j = j0 + tx;
      tile[r][tx] = (k < a.K && j < a.J) ? in[(long long)k * a.J + j] : T(0);
    }
  }
  __syncthreads();
  // rows-of-4: 16 per tile x 64 samples, one 16-B store each
  for (int q = threadIdx.x; q < 16 * 64; q += BLOCK) {
    const int jbl = q >> 6, kl = q & 63;
    const int k = k0 + kl, jb = (j0 >> 2) + jbl;
    if (k < a.K && jb < a.J4) {
      const T v[4] = {tile[kl][4 * jbl + 0], tile[kl][4 * jbl + 1], tile[kl][4 * jbl + 2], tile[kl][4 * jbl + 3]};
      store4<T>(out, a.zp, jb, k, v);
    }
  }
}

// =============================================================================================
// KMPPI interpolation: v_raw[t] = sum_s W[t,s] * clamp(theta[s] + colour(z_S)[s])
// ctrl points of the block's samples live in LDS as [S*NU][BLOCKK] (conflict-free columns).
// `a` describes the support-point stream: a.Tn = S, a.J = S*nu, a.U = theta, a.shift = 0.
// =============================================================================================
template <typename T, int NU, int NOISE>
__global__ void __launch_bounds__(64) kmppi_interp_kernel(const KArgs<T> a, const T* __restrict__ W,
                                                          int Thor, int J4out, T* __restrict__ out) {
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* ctrl = reinterpret_cast<T*>(smem_raw);             // [S*NU][64]
  T* fac = ctrl + (size_t)a.J * 64;                     // [2*NU*NU]
  ActionConsts<T, NU> ac;
  ac.load(a, fac);
  __syncthreads();
  const int lane = threadIdx.x;
  const int kraw = blockIdx.x * 64 + lane;
  const bool active = kraw < a.K;
  const int k = active ? kraw : a.K - 1;
  const int S = a.Tn;
  const int nss = (S + TT - 1) / TT;
  for (int ss = 0; ss < nss; ++ss) {
    T zc[P4 * 4];
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      T r[4];
      noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
      zc[4 * i] = r[0]; zc[4 * i + 1] = r[1]; zc[4 * i + 2] = r[2]; zc[4 * i + 3] = r[3];
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int s = ss * TT + tt;
      if (s < S) {
        T z[NU], v[NU], e[NU];
#pragma unroll
        for (int n = 0; n < NU; ++n) z[n] = zc[tt * NU + n];
        make_action_rt<T, NU>(a, ac, a.U, s, z, -2, v, e);   // theta + eps, clamp (mppi.py:660-663)
#pragma unroll
        for (int n = 0; n < NU; ++n) ctrl[(s * NU + n) * 64 + lane] = v[n];
      }
    }
  }
  // no barrier needed: every lane reads back only its own column
  T ob[4];
  int c = 0;
  long long jb = 0;
  for (int t = 0; t < Thor; ++t) {
    T acc[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) acc[n] = T(0);
    for (int s = 0; s < S; ++s) {
      const T w = W[t * S + s];
#pragma unroll
      for (int n = 0; n < NU; ++n) acc[n] += w * ctrl[(s * NU + n) * 64 + lane];   // :665
    }
#pragma unroll
    for (int n = 0; n < NU; ++n) {
      ob[c++] = acc[n];
      if (c == 4) {
        if (active) {
          T* o = out + (jb * a.zp + k) * 4;
          o[0] = ob[0]; o[1] = ob[1]; o[2] = ob[2]; o[3] = ob[3];
        }
        ++jb; c = 0;
      }
    }
  }
  // flush the padded tail rows (zeros) so that K1's whole-super-step reads see defined data
  while (jb < J4out) {
    for (; c < 4; ++c) ob[c] = T(0);
    if (active) {
      T* o = out + (jb * a.zp + k) * 4;
      o[0] = ob[0]; o[1] = ob[1]; o[2] = ob[2]; o[3] = ob[3];
    }
    ++jb; c = 0;
    ob[0] = ob[1] = ob[2] = ob[3] = T(0);
  }
}

// =============================================================================================
// KMPPI interpolation on the matrix cores (fp32, nu % 4 == 0): the operator is one small GEMM per
// control dimension,  V_n (T x 16 samples) = W (T x S) . Theta'_n (S x 16 samples),
// v_mfma_f32_16x16x4_f32 (an exact fp32 fma chain).  A = W tile (lane (g,c) supplies W[t0+c][4ks+g]),
// B = bounded control points (lane (g,c) supplies support point 4ks+g of sample c: exactly the rows
// that lane loaded from the sample-minor stream, no exchange), D: lane (g,c) holds timesteps
// t0+4g+r of sample c -> the four n of a row-of-4 sit in the same lane and leave as one 16-byte
// store.  One wave = 16 samples x ONE row-of-4 column group q (grid.y): 8 row loads, 32 MFMAs and
// 4 stores per 16-timestep tile, ~64 VGPRs -> 8 waves per SIMD, so the load, MFMA and store phases
// of different waves overlap.  HBM-bound (reads K*S*nu, writes K*T*nu floats); the LDS kernel
// above is the general fallback (any nu, fp64).
// =============================================================================================
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int NU, int NOISE, int SK, bool DIAG>
__global__ void __launch_bounds__(BLOCK) kmppi_interp_mfma_kernel(const KArgs<float> a,
                                                                  const float* __restrict__ W, int Thor,
                                                                  float* __restrict__ out) {
  static_assert(NU % 4 == 0, "rows-of-4 must not straddle timesteps");
  constexpr int P4 = NU / 4;
  constexpr int SP = 4 * SK + 1;            // padded row of the staged operator: conflict-free A fetches
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Wl = reinterpret_cast<float*>(smem_raw);          // [Tpad][SP], zero outside (Thor, S)
  const int Tpad = (Thor + 15) & ~15;
  const int S = a.Tn;
  float* thl = Wl + Tpad * SP;                             // [S*NU] control points theta
  float* fac = thl + S * NU;                               // [2*NU*NU] (full Sigma only)
  const int q = blockIdx.y;
  // per-control constants of this column group (uniform -> scalar loads)
  float sd[4], mu[4], lo[4], hi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = 4 * q + i;
    sd[i] = a.L[n * NU + n]; mu[i] = a.mu[n]; lo[i] = a.umin[n]; hi[i] = a.umax[n];
  }
  ActionConsts<float, NU> ac;
  if constexpr (!DIAG) ac.load(a, fac);
  // the operator goes through LDS, not through vector memory: on gfx950 loads and stores retire
  // in order behind ONE counter, so an A fetch from global would wait for the previous tile's stores
  {
    constexpr int UN = 4;
    const int nW = Tpad * SP;
    for (int base = 0; base < nW; base += BLOCK * UN) {
      float tmp[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = base + u * BLOCK + threadIdx.x;
        const int t = i / SP, sp = i - t * SP;
        const bool ok = i < nW && t < Thor && sp < S;
        tmp[u] = W[ok ? (long long)t * S + sp : 0];
        tmp[u] = ok ? tmp[u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = base + u * BLOCK + threadIdx.x;
        if (i < nW) Wl[i] = tmp[u];
      }
    }
    for (int i = threadIdx.x; i < S * NU; i += BLOCK) thl[i] = a.U[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int g = lane >> 4, c = lane & 15;
  const int kraw = blockIdx.x * 64 + wv * 16 + c;
  const bool active = kraw < a.K;
  const int k = active ? kraw : a.K - 1;

  // B operands: bounded control points theta' = clamp(theta + eps) (mppi.py:660-663)
  float B[SK][4];
  if constexpr (DIAG) {
    float zr[SK][4];
#pragma unroll
    for (int ks = 0; ks < SK; ++ks) {
      const int sp = 4 * ks + g;
      const int sl = sp < S ? sp : S - 1;
      noise4<float, NOISE, 10>(a, (long long)sl * P4 + q, k, zr[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < SK; ++ks) {
      const int sp = 4 * ks + g;
      const int sl = sp < S ? sp : S - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = thl[sl * NU + 4 * q + i] + (zr[ks][i] * sd[i] + mu[i]);    // as make_action<DIAG>
        B[ks][i] = sp < S ? clampT(v, lo[i], hi[i]) : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < SK; ++ks) {
      const int sp = 4 * ks + g;
      const int sl = sp < S ? sp : S - 1;
      float z[NU], v[NU], e[NU];
#pragma unroll
      for (int qq = 0; qq < P4; ++qq) {
        float r[4];
        noise4<float, NOISE, 10>(a, (long long)sl * P4 + qq, k, r);
        z[4 * qq] = r[0]; z[4 * qq + 1] = r[1]; z[4 * qq + 2] = r[2]; z[4 * qq + 3] = r[3];
      }
      make_action<float, NU, false, false>(ac, thl + sl * NU, nullptr, z, -2, v, e);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float b = v[i];
#pragma unroll
        for (int qq = 1; qq < P4; ++qq) b = (qq == q) ? v[4 * qq + i] : b;      // q is wave-uniform
        B[ks][i] = sp < S ? b : 0.f;
      }
    }
  }

  const int nks = (S + 3) / 4;
  for (int t0 = 0; t0 < Thor; t0 += 16) {
    f32x4_t D[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) D[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float* Wr = Wl + (t0 + c) * SP + g;
    float aw[SK];
#pragma unroll
    for (int ks = 0; ks < SK; ++ks) aw[ks] = Wr[4 * ks];     // zero-padded beyond S
#pragma unroll
    for (int ks = 0; ks < SK; ++ks) {
      if (ks < nks) {                       // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
          D[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], B[ks][i], D[i], 0, 0, 0);     // :665
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + 4 * g + r;
      if (active && t < Thor) {
        const float v[4] = {D[0][r], D[1][r], D[2][r], D[3][r]};
        store4<float>(out, a.zp, (long long)t * P4 + q, k, v);
      }
    }
  }
}

// =============================================================================================
// prepare: perturbed_action / noise (K,T,nu) + pert_cost (K)
// =============================================================================================
template <typename T, int NU, int NOISE>
__global__ void __launch_bounds__(BLOCK) prepare_kernel(const KArgs<T> a_in) {
  const KArgs<T> a = env_view(a_in);
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);   // [J] base sequence (noise is added to / measured from)
  T* Un = Ue + a.J;                         // [J] true nominal sequence (action cost)
  T* fac = Un + a.J;
  ActionConsts<T, NU> ac;
  ac.load(a, fac);
  for (int j = threadIdx.x; j < a.J; j += BLOCK) { Ue[j] = u_base(a, j); Un[j] = u_eff(a, j); }
  __syncthreads();
  const int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= a.K) return;
  const int orow = overwrite_row(a, a.k_offset + k);
  const int nss = (a.Tn + TT - 1) / TT;
  T pert = T(0), smooth = T(0), vprev[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) vprev[n] = T(0);
  for (int ss = 0; ss < nss; ++ss) {
    T zc[P4 * 4];
#pragma unroll
    for (int i = 0; i < P4; ++i) {
      T r[4];
      noise4<T, NOISE>(a, (long long)ss * P4 + i, k, r);
      if constexpr (NOISE == MPPI_NOISE_PHILOX) {
        if (a.z != nullptr) store4<T>(const_cast<T*>(a.z), a.zp, (long long)ss * P4 + i, k, r);
      }
      zc[4 * i] = r[0]; zc[4 * i + 1] = r[1]; zc[4 * i + 2] = r[2]; zc[4 * i + 3] = r[3];
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = ss * TT + tt;
      if (t < a.Tn) {
        T z[NU], v[NU], e[NU];
#pragma unroll
        for (int n = 0; n < NU; ++n) z[n] = zc[tt * NU + n];
        make_action_rt<T, NU>(a, ac, Ue, t, z, orow, v, e);
        pert += action_cost_dot_rt<T, NU>(a, ac, Un, t, e);
        if (a.smooth_w != T(0)) {                       // mppi.py:559-562
          T d2 = T(0);
#pragma unroll
          for (int n = 0; n < NU; ++n) { const T d = v[n] - vprev[n]; d2 += d * d; vprev[n] = v[n]; }
          if (t > 0) smooth += d2;
        }
        const long long o = ((long long)k * a.Tn + t) * NU;
#pragma unroll
        for (int n = 0; n < NU; ++n) {
          if (a.pa != nullptr) a.pa[o + n] = v[n];
          if (a.noise != nullptr) a.noise[o + n] = e[n];
        }
      }
    }
  }
  if (a.pert != nullptr) a.pert[k] = pert + a.smooth_w * smooth;
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) cost_block_min_kernel(const KArgs<T> a_in) {
  const KArgs<T> a = env_view(a_in);
  const int k = blockIdx.x * BLOCK + threadIdx.x;
  const T bm = wave_min<T>(k < a.K ? a.cost[k] : inf_v<T>());
  if ((threadIdx.x & (WAVE - 1)) == 0 && k < a.K) a.block_min[k / WAVE] = bm;
}

// =============================================================================================
// K3: weights + per-block partial weighted sums
// grid = (nkc, njt).  Block (kc, jt): samples [kc*BLOCK*R, +BLOCK*R), columns [jt*64, +64).
// Each lane keeps 64 accumulators (one per column), loops its R samples inside each row-of-4 so
// that R independent 1 KiB wave loads are in flight, then one transposing wave reduction.
//   NU == 0 : diagonal Sigma, any nu (element-wise in j; per-column constants in LDS)
//   NU  > 0 : full Sigma (colouring needs whole timesteps; tile = whole super-steps)
// =============================================================================================
template <typename T, int NOISE, int R>
__global__ void __launch_bounds__(BLOCK) weights_partial_diag_kernel(const KArgs<T> a_in) {
  k3_diag_block<T, NOISE, R>(a_in, blockIdx.x, blockIdx.y);
}

// (K,T,nu)-layout variant (MPPI_NOISE_KTN, fp32, diagonal Sigma): the draw is row-major in the
// flat column j, so here a LANE owns 4 columns and walks the samples: every wave load is one
// contiguous 1 KiB of a sample's row, the per-column constants sit in registers, and there is no
// cross-lane reduction at all -- just a fixed-order sum over the 4 waves and the k-chunks.
// grid = (nkc, ceil(J/256), n_env); block (kc, cg): samples [kc*BLOCK*R, +BLOCK*R), columns [cg*256, +256).
template <typename T>
__global__ void __launch_bounds__(BLOCK) weights_partial_ktn_kernel(const KArgs<T> a_in) {
  const KArgs<T> a = env_view(a_in);
  __shared__ T red[BLOCK / WAVE];
  __shared__ T wsum[BLOCK / WAVE][WAVE][4];
  const int kc = blockIdx.x, cg = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int j0 = cg * 256 + lane * 4;
  const bool cols = j0 < a.J;                     // J % 4 == 0: the quad is valid as a whole
  T cU[4], cS[4], cM[4], cLo[4], cHi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = cols ? j0 + c : 0, n = j % a.nu;
    cU[c] = u_base(a, j); cS[c] = a.L[n * a.nu + n]; cM[c] = a.mu[n]; cLo[c] = a.umin[n]; cHi[c] = a.umax[n];
  }
  const T beta = shard_beta(a, red);
  const T inv_lambda = T(1) / a.lambda_;
  const long long n_over = (a.null_action ? 1 : 0) + (long long)a.n_sampler;
  const int per_wave = a.R * BLOCK / (BLOCK / WAVE);            // samples of this chunk per wave
  const int kbeg = kc * a.R * BLOCK + wv * per_wave;
  T acc[4] = {T(0), T(0), T(0), T(0)};
  T eta = T(0);
  constexpr int UN = 8;                       // rows in flight per lane
  for (int i0 = 0; i0 < per_wave; i0 += UN) {
    T zz[UN][4], w[UN];
    int orow[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = kbeg + i0 + u;
      const bool ok = k < a.K;
      const int kq = ok ? k : a.K - 1;
      w[u] = ok ? weight_of<T>(a.cost[kq], beta, inv_lambda) : T(0);
      orow[u] = (ok && a.k_offset + k < n_over) ? overwrite_row(a, a.k_offset + k) : -2;
      // every lane of the wave looks at the same sample: a zero weight skips the row (see the TNK4 K3)
      if (cols && w[u] != T(0)) load4_last<T>(a.z + (long long)kq * a.J + j0, 0, 0, 0, zz[u]);
      else { zz[u][0] = zz[u][1] = zz[u][2] = zz[u][3] = T(0); }
      if (ok && cg == 0 && lane == 0 && a.wnz != nullptr) a.wnz[k] = w[u];
      eta += w[u];
    }
    if (cols) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          T v = cU[c] + (zz[u][c] * cS[c] + cM[c]);
          if (orow[u] == -1) v = T(0);                                               // wave-uniform
          else if (orow[u] >= 0) v = a.sampler[(long long)orow[u] * a.J + j0 + c];
          v = clampT(v, cLo[c], cHi[c]);
          acc[c] += w[u] * (v - cU[c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) wsum[wv][lane][c] = acc[c];
  const T eta_w = eta;                      // identical in every lane of the wave
  __syncthreads();
  if (wv == 0 && cols) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      T s = wsum[0][lane][c];
#pragma unroll
      for (int q = 1; q < BLOCK / WAVE; ++q) s += wsum[q][lane][c];
      a.P_part[(long long)kc * a.Jpad + j0 + c] = s * a.e_scale;
    }
  }
  if (lane == 0) red[wv] = eta_w;
  __syncthreads();
  if (cg == 0 && threadIdx.x == 0) {
    T e = red[0];
#pragma unroll
    for (int q = 1; q < BLOCK / WAVE; ++q) e += red[q];
    a.eta_part[kc] = e;
  }
}

// full-Sigma variant: one tile = SSB super-steps of NU-aligned timesteps, at most 64 columns
template <typename T, int NU, int NOISE>
__global__ void __launch_bounds__(BLOCK) weights_partial_full_kernel(const KArgs<T> a_in) {
  const KArgs<T> a = env_view(a_in);
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT;
  constexpr int SSB = (16 / P4) > 0 ? (16 / P4) : 1;     // super-steps per tile
  constexpr int TJ = SSB * P4 * 4;                      // columns per tile (<= 64)
  static_assert(TJ <= 64, "tile too wide");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);               // [J]
  T* fac = Ue + a.J;                                    // [2*NU*NU]
  __shared__ T red[BLOCK / WAVE];
  __shared__ T wsum[BLOCK / WAVE][64];
  const int kc = blockIdx.x, jt = blockIdx.y;
  ActionConsts<T, NU> ac;
  ac.load(a, fac);
  for (int j = threadIdx.x; j < a.J; j += BLOCK) Ue[j] = u_base(a, j);
  const T beta = shard_beta(a, red);
  const T inv_lambda = T(1) / a.lambda_;

  T acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = T(0);
  T eta = T(0);
  for (int r = 0; r < a.R; ++r) {
    const int k = (kc * a.R + r) * BLOCK + threadIdx.x;
    const bool ok = k < a.K;
    const int kq = ok ? k : a.K - 1;
    const T w = ok ? weight_of<T>(a.cost[kq], beta, inv_lambda) : T(0);
    const int orow = ok ? overwrite_row(a, a.k_offset + k) : -2;
    eta += w;
    if (ok && jt == 0 && a.wnz != nullptr) a.wnz[k] = w;
    if (__ballot(w != T(0)) == 0ull) continue;      // nothing to add from these 64 samples (wave-uniform)
#pragma unroll
    for (int sb = 0; sb < SSB; ++sb) {
      const int ss = jt * SSB + sb;
      if (ss * TT < a.Tn) {
        T zc[P4 * 4];
#pragma unroll
        for (int i = 0; i < P4; ++i) {
          T q[4];
          noise4_last<T, NOISE>(a, (long long)ss * P4 + i, kq, q);
          zc[4 * i] = q[0]; zc[4 * i + 1] = q[1]; zc[4 * i + 2] = q[2]; zc[4 * i + 3] = q[3];
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int t = ss * TT + tt;
          if (t < a.Tn) {
            T z[NU], v[NU], e[NU];
#pragma unroll
            for (int n = 0; n < NU; ++n) z[n] = zc[tt * NU + n];
            make_action_rt<T, NU>(a, ac, Ue, t, z, orow, v, e);
#pragma unroll
            for (int n = 0; n < NU; ++n) acc[(sb * TT + tt) * NU + n] += w * e[n];
          }
        }
      }
    }
  }
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const T colsum = wave_reduce_transpose64<T>(acc);
  wsum[wv][lane] = colsum;
  const T eta_b = block_sum<T>(eta, red);
  if (threadIdx.x < TJ) {
    T s = wsum[0][threadIdx.x];
#pragma unroll
    for (int i = 1; i < BLOCK / WAVE; ++i) s += wsum[i][threadIdx.x];
    const int j = jt * TJ + threadIdx.x;
    if (j < a.Jpad) a.P_part[(long long)kc * a.Jpad + j] = s;
  }
  if (jt == 0 && threadIdx.x == 0) a.eta_part[kc] = eta_b;
}

// =============================================================================================
// K4 finalize / K5 combine
// =============================================================================================
template <typename T>
__device__ __forceinline__ T fixed_sum(const T* __restrict__ p, int n, T* red) {
  // fixed-order: every thread sums a strided slice sequentially, then the block tree
  T s = T(0);
  for (int i = threadIdx.x; i < n; i += BLOCK) s += p[i];
  return block_sum<T>(s, red);
}

// sum over the nkc block partials of column j: 4 waves take every 4th chunk (independent loads,
// 8 in flight), then a fixed-order combine through LDS.  Valid for threadIdx.x < 64 on return.
template <typename T>
__device__ __forceinline__ T column_sum(const KArgs<T>& a, int j, T (*part)[WAVE], int npre = 0, const T* pv = nullptr) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  T s = T(0);
  if (j < a.Jpad) {
    int c = wv;
    if (npre == 16) {                      // the caller loaded this wave's first 16 partials already (same order, same sum)
#pragma unroll
      for (int q = 0; q < 16; ++q) s += pv[q];
      c += 16 * (BLOCK / WAVE);
    }
    for (; c + 7 * (BLOCK / WAVE) < a.nkc; c += 8 * (BLOCK / WAVE)) {
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = a.P_part[(long long)(c + q * (BLOCK / WAVE)) * a.Jpad + j];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; c < a.nkc; c += BLOCK / WAVE) s += a.P_part[(long long)c * a.Jpad + j];
  }
  __syncthreads();
  part[wv][lane] = s;
  __syncthreads();
  T r = part[0][lane];
#pragma unroll
  for (int i = 1; i < BLOCK / WAVE; ++i) r += part[i][lane];
  return r;
}

// grid.x = ceil(J/64) column blocks (+ extra blocks that only write omega)
template <typename T>
__global__ void __launch_bounds__(BLOCK) finalize_kernel(const KArgs<T> a_in, int apply, int ncolblocks) {
  const KArgs<T> a = env_view(a_in);
  __shared__ T red[BLOCK / WAVE];
  __shared__ T part[BLOCK / WAVE][WAVE];
  // the first 16 partials of this wave's column slice are requested BEFORE the two reductions: one memory round trip less
  // in a launch that is a chain of them (column_sum keeps the summation order: the prefetched values are its first terms)
  constexpr int PF = 16, WV = BLOCK / WAVE;
  const int jp = blockIdx.x * WAVE + (threadIdx.x & (WAVE - 1)), wvp = threadIdx.x / WAVE;
  const bool pre = (int)blockIdx.x < ncolblocks && jp < a.Jpad && a.nkc >= PF * WV;
  T pv[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) pv[q] = pre ? a.P_part[(long long)(wvp + q * WV) * a.Jpad + jp] : T(0);
  const T beta = shard_beta(a, red);
  const T eta = fixed_sum<T>(a.eta_part, a.nkc, red);
  const T inv_eta = T(1) / eta;                                        // mppi.py:258
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.record[0] = beta; a.record[1] = eta; }
  if ((int)blockIdx.x < ncolblocks) {
    const int j = blockIdx.x * WAVE + (threadIdx.x & (WAVE - 1));
    const T P = column_sum<T>(a, j, part, pre ? PF : 0, pv);
    if (threadIdx.x < WAVE && j < a.J) {
      a.record[2 + j] = P;
      if (apply) {
        const T un = u_eff(a, j) + P * inv_eta;                        // :268-270
        a.U_out[j] = un;
        if (a.action_out != nullptr && j < a.u_per_command * a.nu) a.action_out[j] = un;   // :271
      }
    }
  }
  if (apply && a.omega != nullptr) {
    const T inv_lambda = T(1) / a.lambda_;
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < a.K; k += gridDim.x * BLOCK)
      a.omega[k] = inv_eta * weight_of<T>(a.cost[k], beta, inv_lambda);
  }
}

// K4 of the on-chip command (rollout_onchip.hpp): the K/256 workgroups left partial records relative to their OWN
// minima -- beta_b = block_min[b], eta_b = eta_part[b], P_b = P_part[b][.] -- combined here in block order with the
// algebra of the multi-GPU combine:  beta = min beta_b,  s_b = exp(-(beta_b - beta)/lambda),  eta = sum s_b eta_b,
// P[j] = sum s_b P_b[j];  then K4 proper: record {beta, eta, P}, U_new = shift(U) + P/eta (mppi.py:258, :268-275).
// grid.x = ceil(J/64) column blocks (+ blocks that only write omega / cost_total_non_zero when those are asked for);
// dynamic LDS: nblk scale factors.
template <typename T>
__global__ void __launch_bounds__(BLOCK) finalize_blocks_kernel(const KArgs<T> a, int apply, int ncolblocks, int nblk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
  T* sb = reinterpret_cast<T*>(fb_smem);       // [nblk]
  __shared__ T red[BLOCK / WAVE];
  __shared__ T part[BLOCK / WAVE][WAVE];
  const T inv_lambda = T(1) / a.lambda_;
  // the first 32 records of this wave's column slice are requested BEFORE the reductions below: their round trip runs under
  // the minimum / eta passes instead of behind them (the launch is a chain of dependent memory round trips and little else)
  constexpr int PF = 32, WV = BLOCK / WAVE;
  const int lane_ = threadIdx.x & (WAVE - 1), wv_ = threadIdx.x / WAVE;
  const int j_ = blockIdx.x * WAVE + lane_;
  const bool cols = (int)blockIdx.x < ncolblocks && j_ < a.J;
  const bool pre = cols && nblk >= PF * WV;
  T pv[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) pv[q] = pre ? a.P_part[(long long)(wv_ + q * WV) * a.Jpad + j_] : T(0);
  T m = inf_v<T>();
  for (int b = threadIdx.x; b < nblk; b += BLOCK) {
    const T v = a.block_min[b];
    sb[b] = v;
    m = v < m ? v : m;
  }
  const T beta = block_min<T>(m, red);         // (its barriers publish sb)
  T es = T(0);
  for (int b = threadIdx.x; b < nblk; b += BLOCK) {
    const T s = m_exp(-inv_lambda * (sb[b] - beta));
    sb[b] = s;
    es += s * a.eta_part[b];
  }
  const T eta = block_sum<T>(es, red);         // fixed order: strided slices, then the block tree
  const T inv_eta = T(1) / eta;                // mppi.py:258
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.record[0] = beta; a.record[1] = eta; }
  if ((int)blockIdx.x < ncolblocks) {
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const int j = blockIdx.x * WAVE + lane;
    T s = T(0);
    if (j < a.J) {
      int c = wv;
      if (pre) {
#pragma unroll
        for (int q = 0; q < PF; ++q) s = m_fma(sb[c + q * WV], pv[q], s);
        c += PF * WV;
      }
      // 32 loads in flight per lane: the records are read once, by 12 workgroups -- nothing but these round trips decides
      // this launch's time (256 records: two batches per wave)
      for (; c + 31 * (BLOCK / WAVE) < nblk; c += 32 * (BLOCK / WAVE)) {
        T v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = a.P_part[(long long)(c + q * (BLOCK / WAVE)) * a.Jpad + j];
#pragma unroll
        for (int q = 0; q < 32; ++q) s = m_fma(sb[c + q * (BLOCK / WAVE)], v[q], s);
      }
      for (; c + 7 * (BLOCK / WAVE) < nblk; c += 8 * (BLOCK / WAVE)) {     // 8 loads in flight per lane
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = a.P_part[(long long)(c + q * (BLOCK / WAVE)) * a.Jpad + j];
#pragma unroll
        for (int q = 0; q < 8; ++q) s = m_fma(sb[c + q * (BLOCK / WAVE)], v[q], s);
      }
      for (; c < nblk; c += BLOCK / WAVE) s = m_fma(sb[c], a.P_part[(long long)c * a.Jpad + j], s);
    }
    part[wv][lane] = s;
    __syncthreads();
    if (threadIdx.x < WAVE && j < a.J) {
      T P = part[0][lane];
#pragma unroll
      for (int i = 1; i < BLOCK / WAVE; ++i) P += part[i][lane];
      a.record[2 + j] = P;
      if (apply) {
        const T un = u_eff(a, j) + P * inv_eta;                            // :268-270
        a.U_out[j] = un;
        if (a.action_out != nullptr && j < a.u_per_command * a.nu) a.action_out[j] = un;   // :271
      }
    }
  }
  if (a.wnz != nullptr || (apply && a.omega != nullptr)) {
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < a.K; k += gridDim.x * BLOCK) {
      const T w = weight_of<T>(a.cost[k], beta, inv_lambda);
      if (a.wnz != nullptr) a.wnz[k] = w;
      if (apply && a.omega != nullptr) a.omega[k] = inv_eta * w;
    }
  }
}

// K5.  The G shard records either sit one behind the other in `rec` (all-gathered: RCCL, copies) or are read where each shard's
// K4 left them (`ptrs`: the device group's staged exchange on one device / between peer-accessible devices, csrc/group.hip) --
// the same arithmetic in the same order either way.
template <typename T> struct RecordPtrs { const T* p[MPPI_MAX_GROUP]; };
template <typename T, bool PTRS>
__global__ void __launch_bounds__(BLOCK) combine_kernel(const KArgs<T> a, const T* __restrict__ rec, const RecordPtrs<T> ptrs,
                                                        int G) {
  const int stride = 2 + a.J;
  auto R = [&](int g, int i) -> T { return PTRS ? ptrs.p[g][i] : rec[(long long)g * stride + i]; };
  T beta = R(0, 0);
  for (int g = 1; g < G; ++g) { const T b = R(g, 0); beta = b < beta ? b : beta; }
  const T inv_lambda = T(1) / a.lambda_;
  T eta = T(0);
  for (int g = 0; g < G; ++g)
    eta += m_exp(-inv_lambda * (R(g, 0) - beta)) * R(g, 1);
  const T inv_eta = T(1) / eta;
  const int gid = blockIdx.x * BLOCK + threadIdx.x;
  if (gid < a.J) {
    T P = T(0);
    for (int g = 0; g < G; ++g)
      P += m_exp(-inv_lambda * (R(g, 0) - beta)) * R(g, 2 + gid);
    const T un = u_eff(a, gid) + P * inv_eta;
    a.U_out[gid] = un;
    if (a.action_out != nullptr && gid < a.u_per_command * a.nu) a.action_out[gid] = un;
  }
  if (a.omega != nullptr) {
    for (int k = gid; k < a.K; k += gridDim.x * BLOCK)
      a.omega[k] = inv_eta * weight_of<T>(a.cost[k], beta, inv_lambda);
  }
}

// =============================================================================================
// host-side launchers
// =============================================================================================
}  // namespace mppi
#include "update_dyn.hpp"
namespace mppi {

// launch one of the runtime-nu kernels (update_dyn.hpp) with its dynamic LDS
#define MPPI_DYN_LAUNCH(KERN, GRID, SMEM, ...)                                                      \
  {                                                                                                 \
    const size_t smem_ = (SMEM);                                                                    \
    if (smem_ > 160 * 1024) return MPPI_E_UNSUPPORTED;                                              \
    if (a.noise_src == MPPI_NOISE_PHILOX) {                                                         \
      if (smem_ > 64 * 1024)                                                                        \
        (void)hipFuncSetAttribute((const void*)KERN<T, MPPI_NOISE_PHILOX>,                          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_);          \
      hipLaunchKernelGGL((KERN<T, MPPI_NOISE_PHILOX>), GRID, dim3(DYN_BLOCK), smem_, st, __VA_ARGS__); \
    } else {                                                                                        \
      if (smem_ > 64 * 1024)                                                                        \
        (void)hipFuncSetAttribute((const void*)KERN<T, MPPI_NOISE_TNK4>,                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_);          \
      hipLaunchKernelGGL((KERN<T, MPPI_NOISE_TNK4>), GRID, dim3(DYN_BLOCK), smem_, st, __VA_ARGS__); \
    }                                                                                               \
    return (int)hipGetLastError();                                                                  \
  }

#define MPPI_NU_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(10) X(12) X(16)

template <typename T>
int launch_noise_fill_philox(const KArgs<T>& a, T* out, hipStream_t st) {
  const dim3 grid((a.K + BLOCK - 1) / BLOCK, a.J4 < 64 ? a.J4 : 64);
  hipLaunchKernelGGL(noise_fill_philox_kernel<T>, grid, dim3(BLOCK), 0, st, a, out);
  return (int)hipGetLastError();
}

template <typename T>
int launch_noise_fill_philox_coloured(const KArgs<T>& a, T* out, hipStream_t st) {
  const int nss = (a.Tn + 3) / 4 + 1;      // >= super-steps for any nu; the kernel strides over the real count
  const dim3 grid((a.K + BLOCK - 1) / BLOCK, nss < 64 ? nss : 64), block(BLOCK);
#define X(N)                                                                                       \
  if (a.nu == N) {                                                                                 \
    hipLaunchKernelGGL((noise_fill_philox_coloured_kernel<T, N>), grid, block, 0, st, a, out);     \
    return (int)hipGetLastError();                                                                 \
  }
  MPPI_NU_LIST(X)
#undef X
  return MPPI_E_UNSUPPORTED;
}

template <typename T>
int launch_noise_from_ktn(const KArgs<T>& a, const T* in, T* out, hipStream_t st) {
  const dim3 grid((a.K + 63) / 64, (a.J4 * 4 + 63) / 64);
  const bool vec = a.J % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL((noise_from_ktn_kernel<T, true>), grid, dim3(BLOCK), 0, st, a, in, out);
  else
    hipLaunchKernelGGL((noise_from_ktn_kernel<T, false>), grid, dim3(BLOCK), 0, st, a, in, out);
  return (int)hipGetLastError();
}

// fp32, nu % 4 == 0, S <= 64: the matrix-core kernel.  Returns 1 when it launched.
template <typename T>
bool try_kmppi_interp_mfma(const KArgs<T>&, const T*, int, int, T*, hipStream_t) { return false; }
template <>
bool try_kmppi_interp_mfma<float>(const KArgs<float>& a, const float* W, int Thor, int J4out, float* out,
                                  hipStream_t st) {
  static const bool off = getenv("MPPI_KMPPI_NO_MFMA") != nullptr;      // A/B knob for tools/, read once
  // (rng="philox7": this kernel generates its rows with the ten-round generator fixed at compile time -- the run-time choice cost
  // its widest instantiation 241 spilled registers -- so the seven-round stream takes the plain interpolation kernel below)
  if (off || a.nu % 4 != 0 || a.Tn > 64 || J4out != Thor * (a.nu / 4) || (a.seven && a.noise_src == MPPI_NOISE_PHILOX)) return false;
  const dim3 grid((a.K + 63) / 64, a.nu / 4), block(BLOCK);
  const int Tpad = (Thor + 15) & ~15;
#define LK2(N, SKK, NS, DG)                                                                         \
  hipLaunchKernelGGL((kmppi_interp_mfma_kernel<N, NS, SKK, DG>), grid, block, smem, st, a, W, Thor, out);
#define LK(N, SKK)                                                                                  \
  {                                                                                                 \
    const size_t smem = ((size_t)Tpad * (4 * SKK + 1) + (size_t)a.Tn * N + 2 * N * N) * sizeof(float); \
    if (smem > 64 * 1024) return false;                                                             \
    if (a.noise_src == MPPI_NOISE_PHILOX) {                                                         \
      if (a.diag) LK2(N, SKK, MPPI_NOISE_PHILOX, true) else LK2(N, SKK, MPPI_NOISE_PHILOX, false)   \
    } else {                                                                                        \
      if (a.diag) LK2(N, SKK, MPPI_NOISE_TNK4, true) else LK2(N, SKK, MPPI_NOISE_TNK4, false)       \
    }                                                                                               \
    return true;                                                                                    \
  }
#define LN(N)                    \
  if (a.nu == N) {               \
    if (a.Tn <= 16) LK(N, 4)     \
    if (a.Tn <= 32) LK(N, 8)     \
    LK(N, 16)                    \
  }
  LN(4) LN(8) LN(12) LN(16)
#undef LN
#undef LK
#undef LK2
  return false;
}

template <typename T>
int launch_kmppi_interp(const KArgs<T>& a, const T* W, int Thor, int J4out, T* out, hipStream_t st) {
  if (a.noise_src == MPPI_NOISE_KTN) return MPPI_E_UNSUPPORTED;
  if (try_kmppi_interp_mfma<T>(a, W, Thor, J4out, out, st)) return (int)hipGetLastError();
  const size_t smem = ((size_t)a.J * 64 + 2 * a.nu * a.nu) * sizeof(T);
  if (smem > 160 * 1024) return MPPI_E_UNSUPPORTED;
  const dim3 grid((a.K + 63) / 64), block(64);
#define X(N)                                                                                      \
  if (a.nu == N) {                                                                                \
    if (a.noise_src == MPPI_NOISE_PHILOX) {                                                       \
      if (smem > 64 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)kmppi_interp_kernel<T, N, MPPI_NOISE_PHILOX>,      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);        \
      hipLaunchKernelGGL((kmppi_interp_kernel<T, N, MPPI_NOISE_PHILOX>), grid, block, smem, st,   \
                         a, W, Thor, J4out, out);                                                 \
    } else {                                                                                      \
      if (smem > 64 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)kmppi_interp_kernel<T, N, MPPI_NOISE_TNK4>,        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);        \
      hipLaunchKernelGGL((kmppi_interp_kernel<T, N, MPPI_NOISE_TNK4>), grid, block, smem, st, a,  \
                         W, Thor, J4out, out);                                                    \
    }                                                                                             \
    return (int)hipGetLastError();                                                                \
  }
  MPPI_NU_LIST(X)
#undef X
  // any other control width: runtime-nu kernel (update_dyn.hpp)
  MPPI_DYN_LAUNCH(kmppi_interp_dyn_kernel, dim3((a.K + DYN_BLOCK - 1) / DYN_BLOCK),
                  ((size_t)a.J + 3 * a.nu) * DYN_BLOCK * sizeof(T), a, W, Thor, J4out, out)
}

template <typename T>
int launch_prepare(const KArgs<T>& a, hipStream_t st) {
  if (a.noise_src == MPPI_NOISE_KTN) return MPPI_E_UNSUPPORTED;   // convert with mppi_noise_from_ktn first
  const size_t smem = ((size_t)2 * a.J + 2 * a.nu * a.nu) * sizeof(T);
  const dim3 grid((a.K + BLOCK - 1) / BLOCK, 1, a.n_env), block(BLOCK);
  if (smem > 160 * 1024) return MPPI_E_UNSUPPORTED;
#define X(N)                                                                                      \
  if (a.nu == N) {                                                                                \
    if (a.noise_src == MPPI_NOISE_PHILOX) {                                                       \
      if (smem > 64 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)prepare_kernel<T, N, MPPI_NOISE_PHILOX>,           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);        \
      hipLaunchKernelGGL((prepare_kernel<T, N, MPPI_NOISE_PHILOX>), grid, block, smem, st, a);    \
    } else {                                                                                      \
      if (smem > 64 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)prepare_kernel<T, N, MPPI_NOISE_TNK4>,             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);        \
      hipLaunchKernelGGL((prepare_kernel<T, N, MPPI_NOISE_TNK4>), grid, block, smem, st, a);      \
    }                                                                                             \
    return (int)hipGetLastError();                                                                \
  }
  MPPI_NU_LIST(X)
#undef X
  MPPI_DYN_LAUNCH(prepare_dyn_kernel, dim3((a.K + DYN_BLOCK - 1) / DYN_BLOCK, 1, a.n_env),
                  ((size_t)2 * a.J + (size_t)4 * a.nu * DYN_BLOCK) * sizeof(T), a)
}

template <typename T>
int launch_cost_block_min(const KArgs<T>& a, hipStream_t st) {
  hipLaunchKernelGGL(cost_block_min_kernel<T>, dim3((a.K + BLOCK - 1) / BLOCK, 1, a.n_env), dim3(BLOCK), 0, st, a);
  return (int)hipGetLastError();
}

template <typename T>
int launch_weights_partial(const KArgs<T>& a, hipStream_t st) {
  if (a.noise_src == MPPI_NOISE_ACTIONS) return MPPI_E_BADARG;
  if (a.noise_src == MPPI_NOISE_KTN) {
    if (!a.diag || a.J % 4 != 0) return MPPI_E_UNSUPPORTED;
    const dim3 grid(a.nkc, (a.J + 255) / 256, a.n_env), block(BLOCK);
    hipLaunchKernelGGL(weights_partial_ktn_kernel<T>, grid, block, 0, st, a);
    return (int)hipGetLastError();
  }
  if (a.diag || a.coloured) {
    const dim3 grid(a.nkc, (a.J4 * 4 + UPD_TJ - 1) / UPD_TJ, a.n_env), block(BLOCK);
#define LAUNCH_R(RR)                                                                              \
  if (a.R == RR) {                                                                                \
    if (a.noise_src == MPPI_NOISE_PHILOX)                                                         \
      hipLaunchKernelGGL((weights_partial_diag_kernel<T, MPPI_NOISE_PHILOX, RR>), grid, block, 0, \
                         st, a);                                                                  \
    else                                                                                          \
      hipLaunchKernelGGL((weights_partial_diag_kernel<T, MPPI_NOISE_TNK4, RR>), grid, block, 0,   \
                         st, a);                                                                  \
    return (int)hipGetLastError();                                                                \
  }
    LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4) LAUNCH_R(8)
#undef LAUNCH_R
    return MPPI_E_BADARG;
  }
  const size_t smem = ((size_t)a.J + 2 * a.nu * a.nu) * sizeof(T);
#define X(N)                                                                                      \
  if (a.nu == N) {                                                                                \
    constexpr int P4 = Stream<N>::P4, TT = Stream<N>::TT;                                         \
    constexpr int SSB = (16 / P4) > 0 ? (16 / P4) : 1;                                            \
    const int nss = (a.Tn + TT - 1) / TT;                                                         \
    const dim3 grid(a.nkc, (nss + SSB - 1) / SSB, a.n_env), block(BLOCK);                                  \
    if (a.noise_src == MPPI_NOISE_PHILOX) {                                                       \
      if (smem > 64 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)weights_partial_full_kernel<T, N, MPPI_NOISE_PHILOX>, \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);        \
      hipLaunchKernelGGL((weights_partial_full_kernel<T, N, MPPI_NOISE_PHILOX>), grid, block,     \
                         smem, st, a);                                                            \
    } else {                                                                                      \
      if (smem > 64 * 1024)                                                                       \
        (void)hipFuncSetAttribute((const void*)weights_partial_full_kernel<T, N, MPPI_NOISE_TNK4>, \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);        \
      hipLaunchKernelGGL((weights_partial_full_kernel<T, N, MPPI_NOISE_TNK4>), grid, block, smem, \
                         st, a);                                                                  \
    }                                                                                             \
    return (int)hipGetLastError();                                                                \
  }
  MPPI_NU_LIST(X)
#undef X
  MPPI_DYN_LAUNCH(weights_partial_full_dyn_kernel, dim3(a.nkc, a.Tn, a.n_env),
                  ((size_t)a.J + (size_t)4 * a.nu * DYN_BLOCK) * sizeof(T), a, a.R * BLOCK)
}

template <typename T>
int launch_finalize(const KArgs<T>& a, int apply, hipStream_t st) {
  const int ncol = (a.J + WAVE - 1) / WAVE;
  int nb = ncol;
  if (apply && a.omega != nullptr) {
    int nbk = (a.K + 4 * BLOCK - 1) / (4 * BLOCK);
    if (nbk > 256) nbk = 256;
    nb = nbk > nb ? nbk : nb;
  }
  hipLaunchKernelGGL(finalize_kernel<T>, dim3(nb, 1, a.n_env), dim3(BLOCK), 0, st, a, apply, ncol);
  return (int)hipGetLastError();
}

template <typename T>
int launch_finalize_blocks(const KArgs<T>& a, int apply, hipStream_t st) {
  if (a.n_env > 1) return MPPI_E_UNSUPPORTED;
  const int ncol = (a.J + WAVE - 1) / WAVE, nblk = a.nkc;
  int nb = ncol;
  if (a.wnz != nullptr || (apply && a.omega != nullptr)) {
    int nbk = (a.K + 4 * BLOCK - 1) / (4 * BLOCK);
    if (nbk > 256) nbk = 256;
    nb = nbk > nb ? nbk : nb;
  }
  hipLaunchKernelGGL(finalize_blocks_kernel<T>, dim3(nb), dim3(BLOCK), (size_t)nblk * sizeof(T), st, a, apply, ncol, nblk);
  return (int)hipGetLastError();
}

template <typename T>
int launch_combine(const KArgs<T>& a, const T* rec, int G, hipStream_t st, const T* const* ptrs) {
  if (a.n_env > 1) return MPPI_E_UNSUPPORTED;   // sharded MPPI_Batched: not built
  int nb = (a.J + BLOCK - 1) / BLOCK;
  if (a.omega != nullptr) {
    const int nbk = (a.K + BLOCK - 1) / BLOCK;
    nb = nbk > nb ? nbk : nb;
    if (nb > 1024) nb = 1024;
  }
  if (ptrs != nullptr) {
    if (G > MPPI_MAX_GROUP) return MPPI_E_UNSUPPORTED;
    RecordPtrs<T> rp;
    for (int g = 0; g < MPPI_MAX_GROUP; ++g) rp.p[g] = g < G ? ptrs[g] : nullptr;
    hipLaunchKernelGGL((combine_kernel<T, true>), dim3(nb), dim3(BLOCK), 0, st, a, (const T*)nullptr, rp, G);
  } else {
    hipLaunchKernelGGL((combine_kernel<T, false>), dim3(nb), dim3(BLOCK), 0, st, a, rec, RecordPtrs<T>{}, G);
  }
  return (int)hipGetLastError();
}

// KMPPI's per-command bookkeeping on the nominal sequences, as ONE tiny launch each instead of the
// roll / copy / two GEMM launches a host-side formulation costs (~5 us of stream time apiece inside a
// ~120 us command):
//   shift        (mppi.py:232-238, :617-619):  theta_out = W_shift theta,  U_out = roll(U, -1), last row = u_init
//   trajectory   (mppi.py:682):                U_out     = W theta
// out[r][n] = sum_s M[r][s] x[s][n] in index order; one thread per output element.
template <typename T>
__global__ void __launch_bounds__(BLOCK) kmppi_sequences_kernel(int R, int S, int nu, const T* __restrict__ M,
                                                                const T* __restrict__ x, T* __restrict__ out, int Troll,
                                                                const T* __restrict__ U, const T* __restrict__ u_init,
                                                                T* __restrict__ U_out) {
  const int g = blockIdx.x * BLOCK + threadIdx.x;
  if (g < R * nu) {
    const int r = g / nu, n = g - r * nu;
    T acc = T(0);
#pragma unroll 8
    for (int s = 0; s < S; ++s) acc = m_fma(M[r * S + s], x[s * nu + n], acc);   // loads of 8 terms in flight, sum in index order
    out[g] = acc;
  } else if (g - R * nu < Troll * nu) {
    const int j = g - R * nu, jn = j + nu;
    U_out[j] = jn < Troll * nu ? U[jn] : u_init[jn - Troll * nu];
  }
}

template <typename T>
int launch_kmppi_sequences(int R, int S, int nu, const T* M, const T* x, T* out, int Troll, const T* U,
                           const T* u_init, T* U_out, hipStream_t st) {
  const int n = (R + Troll) * nu;
  hipLaunchKernelGGL(kmppi_sequences_kernel<T>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, R, S, nu, M, x, out,
                     Troll, U, u_init, U_out);
  return (int)hipGetLastError();
}

// Behind a KMPPI update (ABI 22): the trajectory of the new control points AND both sequences as the next command's shift will want
// them, in one launch -- U = W theta (mppi.py:682); theta_s = W_shift theta (:617-619); U_s = roll(U, -1) with u_init in the last
// row (:232-238), row t of it computed as row t + 1 of W theta with the very fma chain that makes U (the same bits as a roll of U).
// A control loop that shifts every command (the default) then launches nothing for the shift.
template <typename T>
__global__ void __launch_bounds__(BLOCK) kmppi_after_update_kernel(int Tn, int S, int nu, const T* __restrict__ W, const T* __restrict__ Ws,
                                                                   const T* __restrict__ theta, const T* __restrict__ u_init,
                                                                   T* __restrict__ U_out, T* __restrict__ theta_s, T* __restrict__ U_s) {
  const int g = blockIdx.x * BLOCK + threadIdx.x;
  if (g < Tn * nu) {
    const int r = g / nu, n = g - r * nu;
    T acc = T(0);
#pragma unroll 8
    for (int s = 0; s < S; ++s) acc = m_fma(W[r * S + s], theta[s * nu + n], acc);
    U_out[g] = acc;
    if (r > 0) U_s[g - nu] = acc;
    if (r == Tn - 1) U_s[g] = u_init[n];
  } else if (g - Tn * nu < S * nu) {
    const int j = g - Tn * nu, r = j / nu, n = j - r * nu;
    T acc = T(0);
#pragma unroll 8
    for (int s = 0; s < S; ++s) acc = m_fma(Ws[r * S + s], theta[s * nu + n], acc);
    theta_s[j] = acc;
  }
}
template <typename T>
int launch_kmppi_after_update(int Tn, int S, int nu, const T* W, const T* Ws, const T* theta, const T* u_init, T* U_out, T* theta_s,
                              T* U_s, hipStream_t st) {
  const int n = (Tn + S) * nu;
  hipLaunchKernelGGL(kmppi_after_update_kernel<T>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, Tn, S, nu, W, Ws, theta, u_init,
                     U_out, theta_s, U_s);
  return (int)hipGetLastError();
}

// SMPPI.shift_nominal_trajectory (mppi.py:488-492) and the base sequence of the next command (:540) in one
// launch (host-side: two concatenations and an add):
//   U_out = roll(U, -1), last row = u_init;   A_out = roll(A, -1), last row repeats;   B_out = A_out + U_out * dt
template <typename T>
__global__ void __launch_bounds__(BLOCK) smppi_shift_kernel(int J, int nu, const T* __restrict__ U, const T* __restrict__ u_init,
                                                            const T* __restrict__ A, T dt, T* __restrict__ U_out,
                                                            T* __restrict__ A_out, T* __restrict__ B_out) {
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= J) return;
  const int jn = j + nu;
  const T u = jn < J ? U[jn] : u_init[jn - J];
  const T av = jn < J ? A[jn] : A[j];
  U_out[j] = u;
  A_out[j] = av;
  B_out[j] = av + dt * u;
}
template <typename T>
int launch_smppi_shift(int Tn, int nu, const T* U, const T* u_init, const T* A, T dt, T* U_out, T* A_out, T* B_out, hipStream_t st) {
  const int J = Tn * nu;
  hipLaunchKernelGGL(smppi_shift_kernel<T>, dim3((J + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, J, nu, U, u_init, A, dt, U_out, A_out, B_out);
  return (int)hipGetLastError();
}

#define MPPI_INST(T)                                                                     \
  template int launch_noise_fill_philox<T>(const KArgs<T>&, T*, hipStream_t);             \
  template int launch_noise_fill_philox_coloured<T>(const KArgs<T>&, T*, hipStream_t);    \
  template int launch_noise_from_ktn<T>(const KArgs<T>&, const T*, T*, hipStream_t);      \
  template int launch_kmppi_interp<T>(const KArgs<T>&, const T*, int, int, T*, hipStream_t); \
  template int launch_prepare<T>(const KArgs<T>&, hipStream_t);                           \
  template int launch_cost_block_min<T>(const KArgs<T>&, hipStream_t);                    \
  template int launch_weights_partial<T>(const KArgs<T>&, hipStream_t);                   \
  template int launch_finalize<T>(const KArgs<T>&, int, hipStream_t);                     \
  template int launch_finalize_blocks<T>(const KArgs<T>&, int, hipStream_t);              \
  template int launch_combine<T>(const KArgs<T>&, const T*, int, hipStream_t, const T* const*);              \
  template int launch_kmppi_sequences<T>(int, int, int, const T*, const T*, T*, int, const T*, const T*, T*, hipStream_t); \
  template int launch_kmppi_after_update<T>(int, int, int, const T*, const T*, const T*, const T*, T*, T*, T*, hipStream_t); \
  template int launch_smppi_shift<T>(int, int, const T*, const T*, const T*, T, T*, T*, T*, hipStream_t);
MPPI_INST(float)
MPPI_INST(double)

}  // namespace mppi
