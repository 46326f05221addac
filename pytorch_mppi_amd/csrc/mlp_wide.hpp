// mlp_wide.hpp -- dense layers of TRACED models on the matrix cores (VERDICT r03 missing #2).
//
// A traced functor (pytorch_mppi_amd/trace.py) evaluates the user's whole step per lane.  For a dynamics NETWORK that is
// thousands of dependent VALU operations per sample and timestep (the reference's learned pendulum, tests/
// pendulum_approximate.py:47-53: 3 -> 32 -> 32 -> 2 tanh = 1184 multiply-adds), and with one lane per sample K = 8192
// fills 128 of the chip's 1024 SIMDs.  The tracer now keeps `F.linear` layers as layers (mlp_first / mlp_mid /
// mlp_last below, with the activations between them applied to whole register arrays), and this header gives those calls
// two executions:
//
//   WX = false   one lane per sample, every layer a chain of fmas (any kernel of the engine, fp32 and fp64, and -- with
//                plain C++ -- the host check of the trace): the distributed form of a vector IS the vector.
//   WX = true    (rollout_wide_kernel, fp32) SIXTEEN samples per wave: lane (g, j) = (lane >> 4, lane & 15) belongs to
//                sample j, the scalar part of the functor runs replicated in the four lane groups, and a layer is
//                v_mfma_f32_16x16x4_f32 (an exact fp32 fma chain) on the transposed problem  H^T = W . X^T :
//                    A (16 x 4)   lane (g, j) holds W[16 ob + j][feature of k-step, group g]
//                    B (4 x 16)   lane (g, j) holds that feature of sample j
//                    D (16 x 16)  lane (g, j) holds outputs 16 ob + 4 g + r  (r = 0..3) of sample j
//                The D layout of a layer is the B layout of the next one when its weights are fetched in the order
//                (block, r) -> feature 16 block + 4 g + r: hidden activations never leave their registers, no lane
//                exchange between layers.  Only the chain's ends touch the replicated form: the first layer picks its
//                k-step's feature by lane group (two v_cndmask levels), the last one hands every output to all four
//                groups (one ds_bpermute per output).  Weights are loop-invariant reads of the parameter vector: the
//                compiler keeps them in registers across the horizon (26 of them for 3 -> 32 -> 32 -> 2).
//   A workgroup of the wide kernel is four waves = 64 samples = one slot of the per-64-sample cost minima K3 reads.
#pragma once

namespace mppi {

// length of the distributed form of an N-vector
__host__ __device__ constexpr int mlp_dlen(int n, bool wx) { return wx ? ((n + 15) / 16) * 4 : n; }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef float mlp_f32x4 __attribute__((ext_vector_type(4)));

// one of four registers by lane group, as two select levels (operands pinned: left alone the compiler turns the select into
// a dynamically indexed load of the array, i.e. puts it in scratch)
__device__ __forceinline__ float mlp_pick4(float a0, float a1, float a2, float a3, int g) {
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  const float lo = (g & 1) ? a1 : a0, hi = (g & 1) ? a3 : a2;
  return (g & 2) ? hi : lo;
}

// One dense layer of a traced model: y = W x + b, W (OUT x IN) row-major and b (OUT) read from the parameter vector.
//   KIND 0: the layer's input is a replicated vector (the head of a chain, or a layer on its own)
//   KIND 1: its input is the distributed output of the previous layer of the chain
//   PRE:    (wide form) the lane's share of W and b is loaded ONCE, when the model object is built, and stays in registers for
//           the whole horizon -- a load inside the time loop cannot be hoisted by the compiler (it may not speculate it) and
//           would be re-issued every timestep
template <int IN, int OUT, int KIND, bool WX, bool PRE, typename T, typename P>
struct MlpLayer;

// ---- one lane per sample: fma chains (bias first, inputs in index order); weights are wave-uniform reads of the vector ----
template <int IN, int OUT, int KIND, bool PRE, typename T, typename P>
struct MlpLayer<IN, OUT, KIND, false, PRE, T, P> {
  P w, b;
  __device__ __forceinline__ void load(P w_, P b_) { w = w_; b = b_; }
  __device__ __forceinline__ void apply(const T* in, T* out) const {
#pragma unroll 4
    for (int o = 0; o < OUT; ++o) {
      T acc = b != nullptr ? (T)b[o] : T(0);
#pragma unroll
      for (int i = 0; i < IN; ++i) acc = m_fma((T)w[o * IN + i], in[i], acc);
      out[o] = acc;
    }
  }
};

// ---- sixteen samples per wave on the matrix cores (fp32) ----
template <int IN, int OUT, int KIND, bool PRE, typename P>
struct MlpLayer<IN, OUT, KIND, true, PRE, float, P> {
  static constexpr int OB = (OUT + 15) / 16;
  static constexpr int KS = KIND == 0 ? (IN + 3) / 4 : ((IN + 15) / 16) * 4;     // k-steps of four features
  static constexpr int NA = PRE ? OB * KS : 1;
  P w, b;
  float a[NA];            // A operands: a[ob * KS + s] = W[16 ob + j][feature of k-step s for group g]
  mlp_f32x4 c[PRE ? OB : 1];   // accumulator start = bias of the lane's outputs 16 ob + 4 g + r
  // feature multiplied by lane group g in k-step s: replicated input -> 4 s + g; distributed input -> 16 (s / 4) + 4 g + (s % 4)
  static __device__ __forceinline__ int feat(int s, int g) { return KIND == 0 ? 4 * s + g : 16 * (s >> 2) + 4 * g + (s & 3); }
  __device__ __forceinline__ float wval(int ob, int s, int g, int j) const {
    const int o = 16 * ob + j, f = feat(s, g);
    return (o < OUT && f < IN) ? (float)w[(o < OUT ? o : 0) * IN + (f < IN ? f : 0)] : 0.f;
  }
  __device__ __forceinline__ mlp_f32x4 bias4(int ob, int g) const {
    mlp_f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * ob + 4 * g + r;
      v[r] = (b != nullptr && o < OUT) ? (float)b[o < OUT ? o : 0] : 0.f;
    }
    return v;
  }
  __device__ __forceinline__ void load(P w_, P b_) {
    w = w_; b = b_;
    if constexpr (PRE) {
      const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        c[ob] = bias4(ob, g);
#pragma unroll
        for (int s = 0; s < KS; ++s) a[ob * KS + s] = wval(ob, s, g, j);
      }
    }
  }
  // bsrc(s): the B operand of k-step s (this lane's feature of its sample)
  template <class BS>
  __device__ __forceinline__ void run(BS bsrc, float (&d)[mlp_dlen(OUT, true)]) const {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    mlp_f32x4 acc[OB];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) acc[ob] = PRE ? c[ob] : bias4(ob, g);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (KIND == 0 || 16 * (s >> 2) + (s & 3) < IN) {     // (a k-step whose four features all lie beyond IN multiplies zeros)
        const float bv = bsrc(s);
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
          acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(PRE ? a[ob * KS + s] : wval(ob, s, g, j), bv, acc[ob], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[4 * ob + r] = acc[ob][r];
  }
};

// distributed -> replicated: output o = 16 ob + 4 go + r sits in group go's register; every group reads it from there
template <int OUT>
__device__ __forceinline__ void mlp_gather(const float (&d)[mlp_dlen(OUT, true)], float (&out)[OUT]) {
  const int j = threadIdx.x & 15;
#pragma unroll
  for (int o = 0; o < OUT; ++o) out[o] = __shfl(d[4 * (o / 16) + (o & 3)], 16 * ((o & 15) >> 2) + j, 64);
}

// what the generated functors call.  first: replicated -> distributed | mid: distributed -> distributed |
// last: distributed -> replicated | single: replicated -> replicated (a layer on its own)
template <int IN, int OUT, int KIND, bool PRE, typename T, typename P>
__device__ __forceinline__ void mlp_first(const MlpLayer<IN, OUT, KIND, false, PRE, T, P>& l, const T (&in)[IN], T (&d)[OUT]) { l.apply(in, d); }
template <int IN, int OUT, int KIND, bool PRE, typename T, typename P>
__device__ __forceinline__ void mlp_mid(const MlpLayer<IN, OUT, KIND, false, PRE, T, P>& l, const T (&din)[IN], T (&d)[OUT]) { l.apply(din, d); }
template <int IN, int OUT, int KIND, bool PRE, typename T, typename P>
__device__ __forceinline__ void mlp_last(const MlpLayer<IN, OUT, KIND, false, PRE, T, P>& l, const T (&din)[IN], T (&out)[OUT]) { l.apply(din, out); }
template <int IN, int OUT, int KIND, bool PRE, typename T, typename P>
__device__ __forceinline__ void mlp_single(const MlpLayer<IN, OUT, KIND, false, PRE, T, P>& l, const T (&in)[IN], T (&out)[OUT]) { l.apply(in, out); }

template <int IN, int OUT, bool PRE, typename P>
__device__ __forceinline__ void mlp_first(const MlpLayer<IN, OUT, 0, true, PRE, float, P>& l, const float (&in)[IN], float (&d)[mlp_dlen(OUT, true)]) {
  const int g = (threadIdx.x & 63) >> 4;
  l.run([&](int s) {
    const int f = 4 * s;
    return mlp_pick4(in[f < IN ? f : 0], f + 1 < IN ? in[f + 1 < IN ? f + 1 : 0] : 0.f, f + 2 < IN ? in[f + 2 < IN ? f + 2 : 0] : 0.f,
                     f + 3 < IN ? in[f + 3 < IN ? f + 3 : 0] : 0.f, g);
  }, d);
}
template <int IN, int OUT, bool PRE, typename P>
__device__ __forceinline__ void mlp_mid(const MlpLayer<IN, OUT, 1, true, PRE, float, P>& l, const float (&din)[mlp_dlen(IN, true)], float (&d)[mlp_dlen(OUT, true)]) {
  l.run([&](int s) { return din[s]; }, d);          // k-step s = (block s / 4, register s % 4) IS register din[s]
}
template <int IN, int OUT, bool PRE, typename P>
__device__ __forceinline__ void mlp_last(const MlpLayer<IN, OUT, 1, true, PRE, float, P>& l, const float (&din)[mlp_dlen(IN, true)], float (&out)[OUT]) {
  float d[mlp_dlen(OUT, true)];
  l.run([&](int s) { return din[s]; }, d);
  mlp_gather<OUT>(d, out);
}
template <int IN, int OUT, bool PRE, typename P>
__device__ __forceinline__ void mlp_single(const MlpLayer<IN, OUT, 0, true, PRE, float, P>& l, const float (&in)[IN], float (&out)[OUT]) {
  float d[mlp_dlen(OUT, true)];
  mlp_first(l, in, d);
  mlp_gather<OUT>(d, out);
}

// a model opts in with `static constexpr bool WIDE = true` and `typedef ... Wide` (its WX = true instantiation)
template <class M, typename = void>
struct model_wide : std::false_type {};
template <class M>
struct model_wide<M, std::enable_if_t<M::WIDE>> : std::true_type {};

constexpr int WIDE_SAMPLES = 16;                       // samples per wave
constexpr int WIDE_WG_SAMPLES = 64;                    // per 256-thread workgroup = one slot of block_min (K3 reads one per 64)

// K1 for a wide model: the rolled ("heavy") stream of rollout.hpp, sixteen samples per wave.  Lane (g, j): sample
// 64 chunk + 16 wave + j; the four groups run the same scalar code on the same rows, group 0 stores.
template <class Model, int NOISE, bool DIAG>
__global__ void __launch_bounds__(K1_BLOCK) rollout_wide_kernel(const KArgs<float> a_in) {
  using T = float;
  constexpr int NX = Model::NX, NU = Model::NU;
  static_assert(K1_BLOCK == 256, "four waves of sixteen samples per workgroup");
  const KArgs<T> a = a_in;
  stamp_entry(a.tstamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);   // [J] nominal sequence, shift applied
  T* Um = Ue + a.J;                         // [J] Ue + mu
  T* G = Um + a.J;                          // [J] lambda * Sigma^-1 Ue[t]
  T* red = G + a.J;                         // [BLOCK/WAVE]
  T* fac = red + BLOCK / WAVE;              // [2*NU*NU] full-Sigma factors (only if !DIAG)
  ActionConsts<T, NU> ac;
  ac.load(a, DIAG ? nullptr : fac);
  for (int j = threadIdx.x; j < a.J; j += K1_BLOCK) {
    const int n = j % NU;
    const T ub = u_base(a, j);
    Ue[j] = ub;
    Um[j] = a.coloured ? ub : ub + a.mu[n];
    if constexpr (DIAG) G[j] = a.lambda_ * (u_eff(a, j) * a.sinv[n * NU + n]);
  }
  const bool coloured_full = DIAG && a.coloured && !a.diag;
  if (coloured_full) {
    for (int i = threadIdx.x; i < NU * NU; i += K1_BLOCK) fac[i] = a.sinv[i];
  }
  __syncthreads();
  if (coloured_full || !DIAG) {
    const T* Sinv = coloured_full ? fac : ac.Sm;
    for (int j = threadIdx.x; j < a.J; j += K1_BLOCK) {
      const int n = j % NU, t0 = j - n;
      T g = T(0);
      for (int m = 0; m < NU; ++m) g = m_fma(Sinv[n * NU + m], u_eff(a, t0 + m), g);
      G[j] = a.lambda_ * g;
    }
    __syncthreads();
  }
  const Model model(a);
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE, grp = lane >> 4, jj = lane & 15;
  const int nchunks = (a.K + WIDE_WG_SAMPLES - 1) / WIDE_WG_SAMPLES;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int kraw = chunk * WIDE_WG_SAMPLES + WIDE_SAMPLES * wv + jj;
    const bool inside = kraw < a.K;
    const bool active = inside && grp == 0;           // the lane that stores for its sample
    const int k = inside ? kraw : a.K - 1;            // tail lanes shadow the last sample
    const int orow = overwrite_row(a, a.k_offset + k);
    StepTables<T> tb{Ue, Um, G, nullptr, chunk * WIDE_WG_SAMPLES + WIDE_SAMPLES * wv};
    T x[NX];
    {
      const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = s0[i];      // mppi.py:302-305
    }
    T rollout = T(0), pert = T(0);
    rollout_stream_heavy<Model, T, NOISE, DIAG>(a, ac, model, tb, k, active, orow, x, rollout, pert);
    if (a.use_terminal) rollout += model.terminal(x);                  // :324-328
    const T total = rollout + pert;                                    // :416
    if (active) {
      a.cost[k] = total;
      if (a.pert != nullptr) a.pert[k] = pert;
    }
    const T bm = block_min<T>(inside ? total : inf_v<T>(), red);       // one minimum per 64 samples = per workgroup
    if (threadIdx.x == 0) a.block_min[chunk] = bm;
  }
  if (a.tstamp != nullptr) {
    __syncthreads();
    stamp_exit(a.tstamp);
  }
}

// -1: not this path (the caller goes on with the one-lane-per-sample kernels)
template <class Model, typename T>
static int launch_rollout_wide(const KArgs<T>& a_in, hipStream_t st) {
  if constexpr (!std::is_same<T, float>::value || !model_wide<Model>::value) {
    return -1;
  } else {
    using WM = typename Model::Wide;
    constexpr int NU = Model::NU;
    static const int off = [] { const char* e = getenv("MPPI_WIDE"); return e ? atoi(e) == 0 : 0; }();
    if (off || (a_in.model_flags & MPPI_MODEL_FLAG_NO_WIDE) != 0 || a_in.M != 1 || a_in.n_env != 1 || a_in.W != nullptr || a_in.noise_src == MPPI_NOISE_KTN || a_in.B != nullptr ||
        a_in.smooth_w != 0.f || a_in.nb1 != (a_in.K + WAVE - 1) / WAVE)
      return -1;
    KArgs<T> a = a_in;
    const bool diag = a.diag != 0 || a.coloured != 0;
    const size_t smem = (size_t)(3 * a.J + BLOCK / WAVE + (a.diag != 0 ? 0 : 2 * NU * NU)) * sizeof(T);
    if (smem > 160 * 1024) return -1;
    const int nchunks = (a.K + WIDE_WG_SAMPLES - 1) / WIDE_WG_SAMPLES;
    const dim3 grid(nchunks < 4096 ? nchunks : 4096), block(K1_BLOCK);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    profile_next_events(&ev0, &ev1, &a.tstamp);
#define MPPI_WIDE_LAUNCH1(KERNEL)                                                                                 \
  do {                                                                                                            \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    if (ev1 != nullptr) hipExtLaunchKernelGGL(KERNEL, grid, block, smem, st, ev0, ev1, 0, a);                     \
    else hipLaunchKernelGGL(KERNEL, grid, block, smem, st, a);                                                    \
  } while (0)
#define MPPI_WIDE_LAUNCH(NOISE_)                                                  \
  do {                                                                            \
    if (diag) MPPI_WIDE_LAUNCH1((rollout_wide_kernel<WM, NOISE_, true>));         \
    else MPPI_WIDE_LAUNCH1((rollout_wide_kernel<WM, NOISE_, false>));             \
  } while (0)
    if (a.noise_src == MPPI_NOISE_PHILOX) MPPI_WIDE_LAUNCH(MPPI_NOISE_PHILOX);
    else if (a.noise_src == MPPI_NOISE_ACTIONS) MPPI_WIDE_LAUNCH(MPPI_NOISE_ACTIONS);
    else MPPI_WIDE_LAUNCH(MPPI_NOISE_TNK4);
#undef MPPI_WIDE_LAUNCH
#undef MPPI_WIDE_LAUNCH1
    return (int)hipGetLastError();
  }
}
#endif  // device / hipcc

}  // namespace mppi
