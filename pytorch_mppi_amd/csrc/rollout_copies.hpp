// rollout_copies.hpp -- M > 1 state rollouts per action sequence (mppi.py:334-373), ONE WAVE PER ROLLOUT COPY (round 6).
// Included by rollout.hpp (needs Stream, StepTables, make_action, the models).
//
// rollout_stream_multi (rollout.hpp) keeps the M copies of a sample's state in ONE lane and walks them one after the other: per
// timestep one action, then M x (dynamics step + ceil(nx/4) Philox blocks of process noise + cost).  At C3 with M = 3 that is 768
// Philox4x32-10 + Box-Muller evaluations per sample against the plain command's 192, issued by ONE wave per SIMD (K = 65536 IS one
// wave per SIMD of lanes): 257.6 us, VALU-bound with nothing to fill the gaps between dependent multiplies (profiles/
// r06_a_trace_philox_m3.txt).  Here a workgroup is 64 samples x 4 waves and wave m rolls out copy m of its 64 samples: the same
// arithmetic per copy -- the same process-noise counters ((t MM + m) ceil(nx/4) + block: the same stream), the same order of every
// sum -- spread over four times the waves, so that four independent instruction streams share each SIMD.  What the copies owe each
// other is one number per timestep -- the cost c_m(t), whose variance across m enters the total (:363-364) -- exchanged through LDS
// in chunks of 8 timesteps (double-buffered: one barrier per chunk); every wave then accumulates the discounted variance itself.
// Wave 0 adds the copies' totals in copy order and writes the sample's cost.  Bit for bit the result of rollout_stream_multi
// (tests/test_gpu_multi_copies.py).  The action rows are read by every wave of the workgroup (M reads of a row out of L2 for one out
// of HBM).  Scope: diagonal (or generator-coloured) Sigma, rows in memory (standard normals or KMPPI's raw actions), no sampler
// rows (the null-action row is handled), no `states` output, one environment, light models; anything else keeps
// rollout_stream_multi.  MPPI_MULTI_COPIES=0 forces that form (A/B).
#pragma once

namespace mppi {

constexpr int COPIES_CH = 8;                 // timesteps per exchange of the copies' costs

template <class Model, typename T, int NOISE>
__global__ void __launch_bounds__(WAVE * PROCESS_NOISE_MM, (sizeof(T) == 4 ? 4 : 1)) rollout_copies_kernel(const KArgs<T> a) {
  constexpr int NX = Model::NX, NU = Model::NU, MM = PROCESS_NOISE_MM, CH = COPIES_CH;
  constexpr int P4 = Stream<NU>::P4, TT = Stream<NU>::TT, NXB = (NX + 3) / 4;
  stamp_entry(a.tstamp);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ue = reinterpret_cast<T*>(smem_raw);   // [J] base sequence (shift applied; SMPPI: A + U dt)
  T* G = Ue + a.J;                          // [J] lambda * (Sigma^-1 U[t])[n]
  T* cbuf = G + a.J;                        // [2][CH][MM][WAVE] the copies' step costs of two chunks
  T* totbuf = cbuf + 2 * CH * MM * WAVE;    // [MM][WAVE] the copies' summed costs (+ terminal)
  T* Csd = totbuf + MM * WAVE;              // [NU] sqrt(diag Sigma) | [NU] mu | [NU] lower bound | [NU] upper bound: out of LDS, not
  T* Cmu = Csd + NU, *Clo = Cmu + NU, *Chi = Clo + NU;   // registers -- four waves per SIMD leave a lane 128 of them
  const int lane = threadIdx.x & (WAVE - 1), m = threadIdx.x / WAVE;
  const int M = a.M;
  const bool mine = m < M;                  // (M = 3: the fourth wave only keeps the barriers company)

  const Model model(a);
  if (threadIdx.x < NU) {
    const int n = threadIdx.x;
    Csd[n] = a.coloured ? T(1) : a.L[n * NU + n];        // (coloured stream: eps is in z already -- ActionConsts::load)
    Cmu[n] = a.coloured ? T(0) : a.mu[n];
    Clo[n] = a.umin[n];
    Chi[n] = a.umax[n];
  }
  const T e_scale = a.e_scale;
  const bool abs_cost = a.abs_cost != 0;
  const bool coloured_full = a.coloured && !a.diag;      // generator-coloured rows of a full Sigma: G needs the whole row
  for (int j = threadIdx.x; j < a.J; j += blockDim.x) Ue[j] = u_base(a, j);
  __syncthreads();
  for (int j = threadIdx.x; j < a.J; j += blockDim.x) {
    const int n = j % NU, t0 = j - n;
    T g;
    if (coloured_full) {
      g = T(0);
#pragma unroll
      for (int q = 0; q < NU; ++q) g = m_fma(a.sinv[n * NU + q], a.B != nullptr ? u_eff(a, t0 + q) : Ue[t0 + q], g);
    } else {
      g = (a.B != nullptr ? u_eff(a, j) : Ue[j]) * a.sinv[n * NU + n];
    }
    G[j] = a.lambda_ * g;
  }
  __syncthreads();

  const T inv_M = T(1) / (T)M, inv_Mm1 = T(1) / (T)(M - 1);
  const int nchunks = (a.K + WAVE - 1) / WAVE;
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int kraw = chunk * WAVE + lane;
    const bool active = kraw < a.K;
    const int k = active ? kraw : a.K - 1;
    const int orow = overwrite_row(a, a.k_offset + k);          // -1 (the null-action row) or -2: no sampler rows here
    T x[NX];
    {
      const T* __restrict__ s0 = a.state_per_sample ? a.state + (long long)k * NX : a.state;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = s0[i];
    }
    T cs = T(0), cvar = T(0), dpow = T(1), pert = T(0), smooth = T(0), vprev[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) vprev[n] = T(0);
    int buf = 0, t_done = 0;                 // cbuf half in use; timesteps whose variance has been accumulated
    const int nss = (a.Tn + TT - 1) / TT;
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
        const int t = ss * TT + tt;
        if (t >= a.Tn) break;
        // ---- the action of this step: every wave of the workgroup makes it for itself -- make_action<DIAG> + the cost terms of
        // rollout_stream_multi, one control dimension at a time (same arithmetic, same order of every sum) ----
        T u[NU], d2 = T(0);
#pragma unroll
        for (int n = 0; n < NU; ++n) {
          const T zz = zc[tt * NU + n], Utn = Ue[t * NU + n];
          T v = NOISE == MPPI_NOISE_ACTIONS ? zz : Utn + (zz * Csd[n] + Cmu[n]);
          if (orow == -1) v = T(0);                                               // the sample_null_action row (mppi.py:390-392)
          v = clampT(v, Clo[n], Chi[n]);
          const T e = (v - Utn) * e_scale;                                        // e_scale = 1 | 1/dt (SMPPI, :544)
          if (a.smooth_w != T(0)) {
            const T d = v - vprev[n];
            d2 = m_fma(d, d, d2);
            vprev[n] = v;
          }
          pert = m_fma(G[t * NU + n], abs_cost ? m_abs(e) : e, pert);             // :409, :415
          u[n] = a.u_scale * v;                                                   // :354
        }
        if (a.smooth_w != T(0) && t > 0) smooth = m_fma(a.smooth_w, d2, smooth);
        // ---- this wave's copy of the state ----
        T c = T(0);
        if (mine) {
          model.step(x, u, t);                                                    // :356
          if (a.proc_sd != nullptr) {
#pragma unroll
            for (int q = 0; q < NXB; ++q) {
              T w[4];
              philox_normal4<T>(a.seed ^ PROCESS_NOISE_KEY_TAG, a.call, a.k_offset + k, ((long long)t * MM + m) * NXB + q, w, a.seven != 0);
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (4 * q + i < NX) x[4 * q + i] = m_fma(a.proc_sd[4 * q + i], w[i], x[4 * q + i]);
            }
          }
          c = model.cost(x, u, t);                                                // :361
          cs += c;                                                                // :362
        }
        cbuf[((buf * CH + (t - t_done)) * MM + m) * WAVE + lane] = c;
        // ---- a chunk of step costs is complete: every wave accumulates the discounted variance across the copies ----
        if (t - t_done == CH - 1 || t == a.Tn - 1) {
          __syncthreads();
          const int nt = t - t_done + 1;
          for (int q = 0; q < nt; ++q) {
            T cm[MM], mean = T(0);
#pragma unroll
            for (int mm = 0; mm < MM; ++mm) {
              cm[mm] = cbuf[((buf * CH + q) * MM + mm) * WAVE + lane];            // (copies >= M wrote 0)
              mean += cm[mm];
            }
            mean *= inv_M;
            T var = T(0);
#pragma unroll
            for (int mm = 0; mm < MM; ++mm) {
              const T d = cm[mm] - mean;
              if (mm < M) var = m_fma(d, d, var);
            }
            cvar = m_fma(var * inv_Mm1, dpow, cvar);                              // :363-364
            dpow *= a.var_disc;
          }
          t_done = t + 1;
          buf ^= 1;     // the next chunk goes to the other half: nobody writes this one again before everybody has passed
                        // the NEXT barrier, i.e. has finished reading it
        }
      }
    }
    // ---- the copies' totals meet: wave 0 adds them in copy order (:369-372) ----
    totbuf[m * WAVE + lane] = mine ? cs + (a.use_terminal ? model.terminal(x) : T(0)) : T(0);
    __syncthreads();
    if (m == 0) {
      T tot = T(0);
#pragma unroll
      for (int mm = 0; mm < MM; ++mm)
        if (mm < M) tot += totbuf[mm * WAVE + lane];
      const T rollout = tot * inv_M + a.var_cost * cvar + smooth;                 // (+ SMPPI :561)
      const T total = rollout + pert;                                             // :416
      if (active) {
        a.cost[k] = total;
        if (a.pert != nullptr) a.pert[k] = pert;
      }
      const T bm = wave_min<T>(active ? total : inf_v<T>());
      if (lane == 0) a.block_min[chunk] = bm;                                     // one minimum per 64 samples
    }
    __syncthreads();                          // totbuf / cbuf are reused by the next chunk of this workgroup
  }
  if (a.tstamp != nullptr) {
    __syncthreads();
    stamp_exit(a.tstamp);
  }
}

// -1: not this form (the caller launches rollout_stream_multi); else a HIP error code (0 = launched)
template <class Model, typename T>
static int launch_rollout_copies(const KArgs<T>& a_in, hipStream_t st) {
  const char* knob = getenv("MPPI_MULTI_COPIES");           // "0": rollout_stream_multi (A/B, tests; read per launch)
  const bool off = knob != nullptr && knob[0] == '0';
  KArgs<T> a = a_in;
  const bool diag = a.diag != 0 || a.coloured != 0;
  if (off || a.M < 2 || a.M > PROCESS_NOISE_MM || !diag || a.n_sampler > 0 || a.states != nullptr || a.n_env > 1 ||
      (a.noise_src != MPPI_NOISE_TNK4 && a.noise_src != MPPI_NOISE_ACTIONS))
    return -1;
  const size_t smem = (size_t)(2 * a.J + 2 * COPIES_CH * PROCESS_NOISE_MM * WAVE + PROCESS_NOISE_MM * WAVE + 4 * Model::NU) * sizeof(T);
  if (smem > 64 * 1024) return -1;
  const int nchunks = (a.K + WAVE - 1) / WAVE;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int gx = nchunks < 8 * n_cu ? nchunks : 8 * n_cu;        // persistent beyond eight workgroups per CU
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  profile_next_events(&ev0, &ev1, &a.tstamp);
  const dim3 grid(gx), block(WAVE * PROCESS_NOISE_MM);
  if (a.noise_src == MPPI_NOISE_ACTIONS) {
    if (ev1 != nullptr) hipExtLaunchKernelGGL((rollout_copies_kernel<Model, T, MPPI_NOISE_ACTIONS>), grid, block, smem, st, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((rollout_copies_kernel<Model, T, MPPI_NOISE_ACTIONS>), grid, block, smem, st, a);
  } else {
    if (ev1 != nullptr) hipExtLaunchKernelGGL((rollout_copies_kernel<Model, T, MPPI_NOISE_TNK4>), grid, block, smem, st, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((rollout_copies_kernel<Model, T, MPPI_NOISE_TNK4>), grid, block, smem, st, a);
  }
  return (int)hipGetLastError();
}

}  // namespace mppi
