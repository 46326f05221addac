/* mppi_amd.h -- C-ABI of the MI355X-native MPPI rollout-and-update engine.
 *
 * The reference (UM-ARM-Lab/pytorch_mppi v0.9.1) has no FFI of its own: its hot path is the
 * private-method chain of `pytorch_mppi.MPPI` in src/pytorch_mppi/mppi.py.  Each entry point
 * below names the reference lines it replaces.  Every function
 *   - is `extern "C"`, takes plain pointers/sizes (device pointers are raw HIP addresses),
 *   - launches on the HIP stream it is handed and never synchronises the device,
 *   - allocates nothing (all workspace is caller-provided, see mppi_workspace_elems; the measurement hook owns its event pool),
 *   - returns 0 on success, a negative MPPI_E_* code for an engine-side refusal, or a
 *     positive hipError_t value; mppi_last_error() returns a thread-local message.
 *
 * All arrays hold `dtype` elements (MPPI_F32 / MPPI_F64 = the dtype of `noise_sigma`,
 * mppi.py:88).  Names follow the reference: K samples, T horizon, nx/nu state/control dims,
 * U nominal control sequence, S KMPPI support points.
 *
 * Noise layouts
 *   MPPI_NOISE_TNK4   engine-native, sample-minor: z4[jb][k][c] with j = 4*jb + c the flat
 *                     (t*nu + n) index, jb < J4 = mppi_noise_rows4(T, nu), k < K <= noise_pitch
 *                     (the row pitch: J4 * noise_pitch * 4 elements in all).  One wave reads
 *                     1 KiB contiguous per instruction (16 B per lane).
 *   MPPI_NOISE_PHILOX no array: the float4 at [jb][k] is generated in-kernel from
 *                     Philox4x32-10(counter = (k_global, jb, call_lo, call_hi), key = seed)
 *                     + Box-Muller, so results do not depend on launch geometry or on the
 *                     number of shards.  If p->z is non-NULL in this mode, mppi_rollout_cost /
 *                     mppi_prepare also STORE the rows they generate there (TNK4), so that the
 *                     update pass can re-read them (noise_src = TNK4) instead of regenerating:
 *                     one Philox pass per command, identical numbers either way.
 * The reference's own layout (K,T,nu) (mppi.py:203) is converted with mppi_noise_from_ktn.
 *
 * KMPPI (mppi.py:593-688): mppi_kmppi_interp turns support-point noise (K,S,nu) into raw
 * trajectories W*clamp(theta+eps) in TNK4 layout, which K1 consumes as MPPI_NOISE_ACTIONS;
 * the theta update is K3/K4 run on a problem whose "sequence" is theta (T:=S, U:=theta).
 */
#ifndef MPPI_AMD_H
#define MPPI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_ABI_VERSION 22

enum { MPPI_F32 = 0, MPPI_F64 = 1 };
enum { MPPI_NEXT_DRAW_TORCH = 0, MPPI_NEXT_DRAW_PHILOX = 1 };   /* MppiProblem.next_kind (ABI 21) */
enum { MPPI_NOISE_TNK4 = 0, MPPI_NOISE_PHILOX = 1,
       MPPI_NOISE_ACTIONS = 2 /* p->z holds pre-made raw actions (TNK4), KMPPI; K1/prepare only */,
       MPPI_NOISE_KTN = 3     /* p->z is the reference's own (K,T,nu) row-major fp32 draw (mppi.py:203), read in
                                 place by mppi_rollout_cost (LDS-transposed 128-B lines) and mppi_weights_partial
                                 (lane = column).  fp32, diagonal Sigma, T*nu % 4 == 0, nu in {4,8,12,16};
                                 anything else returns MPPI_E_UNSUPPORTED -> convert with mppi_noise_from_ktn */ };

/* native dynamics/cost models (device functors in pytorch_mppi_amd/csrc/models.hpp) */
enum {
  MPPI_MODEL_NONE = 0,        /* no fused model: generic path, callbacks stay in Python   */
  MPPI_MODEL_PENDULUM = 1,    /* reference tests/pendulum.py:30-60                        */
  MPPI_MODEL_INTEGRATOR = 2,  /* "quad-toy": reference tests/benchmark_mppi.py:65-78      */
  MPPI_MODEL_LINEAR_GOAL = 3, /* reference tests/test_mppi.py:25-51                       */
  MPPI_MODEL_MLP = 4,         /* x + s*(W2 tanh(W1[x;u]+b1)+b2), tests/pendulum_approximate.py:47-67 */
  MPPI_MODEL_CUSTOM_BASE = 100 /* 100 + slot: user dynamics/cost compiled at run time (jit.py)  */
};

enum {
  MPPI_E_BADARG = -1,      /* null pointer / inconsistent sizes                            */
  MPPI_E_UNSUPPORTED = -2, /* no kernel instantiated for this (model, nx, nu, dtype, ...)  */
  MPPI_E_WORKSPACE = -3,   /* workspace too small                                          */
  MPPI_E_DIST = -4,        /* RCCL reported an error (message in mppi_last_error)          */
  MPPI_E_GROUP_PEER = -5   /* device group (ABI 22): ANOTHER device failed to issue its part of the command (or the
                              command was abandoned); this device issued no exchange and no combine */
};

/* One command()'s worth of inputs/outputs.  Pointers marked [opt] may be NULL. */
/* model_flags */
#define MPPI_MODEL_FLAG_EXACT_FP32 1   /* MPPI_MODEL_MLP: some |W2| >= 3e4 lies outside the fp16 operand range of the split
                                          matrix-core kernel -> run the exact fp32 MFMA kernel (no range limit) instead */
#define MPPI_MODEL_FLAG_NO_WIDE 2      /* run-time registered models with dense layers (csrc/mlp_wide.hpp): keep the one-lane-per-sample
                                          kernels (fma chains) instead of the matrix-core kernel of sixteen samples per wave (A/B, tests) */

typedef struct MppiProblem {
  /* ---- dimensions ---- */
  int32_t K;              /* samples held by THIS shard                                    */
  int32_t T, nx, nu;
  int32_t S;              /* KMPPI support points; 0 for plain MPPI                        */
  int32_t dtype;          /* MPPI_F32 | MPPI_F64                                           */
  int64_t k_offset;       /* global index of this shard's sample 0 (row bookkeeping, RNG)  */
  /* ---- configuration ---- */
  int32_t model_id;
  int32_t sigma_diagonal;     /* mppi.py:131                                               */
  int32_t noise_abs_cost;     /* mppi.py:190,196                                           */
  int32_t sample_null_action; /* mppi.py:390-392: global row 0 := 0                        */
  int32_t n_sampler_rows;     /* mppi.py:393-399: rows [null, null+n) := sampler_actions   */
  int32_t state_per_sample;   /* state is (K,nx) instead of (nx,)  (mppi.py:302-305)       */
  int32_t shift;              /* apply shift_nominal_trajectory (mppi.py:232-238) on read  */
  int32_t use_terminal;       /* add the model's terminal cost (mppi.py:324-328)           */
  int32_t noise_src;          /* MPPI_NOISE_*                                              */
  int32_t u_per_command;      /* mppi.py:271                                               */
  int32_t rollout_samples;    /* M (mppi.py:76, :334-373): state rollouts per action sequence (fused: <= 4); 0/1 = one */
  int32_t hidden;             /* MLP hidden width                                          */
  int32_t num_envs;           /* MPPI_Batched (mppi.py:691-873): N independent controllers that
                                 share ONE noise draw; 0/1 = single.  state (N,nx), U / U_out
                                 (N,T,nu), cost_total / omega / cost_total_non_zero (N,K), record
                                 (N,2+J), perturbed_action / noise (N,K,T,nu), pert_cost (N,K);
                                 the environment is the z axis of every launch grid          */
  int32_t noise_coloured;     /* p->z already holds eps = L z + mu (mppi_noise_fill_philox_coloured): kernels add U and
                               * bound only; the action cost still uses the true Sigma^-1 */
  int32_t model_flags;        /* MPPI_MODEL_FLAG_*: per-model kernel choices the host knows about its parameters (ABI 18) */
  double lambda_;             /* mppi.py:96, read live                                     */
  double u_scale;             /* mppi.py:313                                               */
  uint64_t seed, call;        /* Philox key / per-command counter word                     */
  int64_t noise_pitch;        /* TNK4 row pitch in samples: row jb starts at element jb*noise_pitch*4;
                                 0 = K (dense).  mppi_noise_pitch(K, dtype) is the engine's choice   */
  /* SMPPI (mppi.py:451-570), lifted control: v = clamp(A + (U+eps)*dt), noise = (v-A)/dt - U.
   * The host passes base_seq = A + U*dt, noise_L/noise_mu pre-multiplied by dt and the ACTION
   * bounds in u_min/u_max; the kernels then only need 1/dt and the smoothness weight.         */
  double noise_rescale;       /* 1/dt (SMPPI) -- multiplies the bounded noise; 1 for MPPI   */
  double smooth_weight;       /* w_action_seq_cost * u_scale^2 (mppi.py:559-562); 0 = off   */
  double rollout_var_cost;    /* M > 1: weight of the discounted cost variance over the M rollouts (mppi.py:77, :372) */
  double rollout_var_discount;/* M > 1: its per-step discount (mppi.py:78, :174-175, :364)  */
  /* ---- inputs (device) ---- */
  const void* state;          /* (nx) or (K,nx)                                            */
  const void* U;              /* (T,nu) nominal sequence BEFORE this command's shift       */
  const void* u_init;         /* (nu)                                                      */
  const void* noise_mu;       /* (nu)                                                      */
  const void* noise_L;        /* (nu,nu) row-major: chol(Sigma), or diag(sqrt(diag))       */
  const void* sigma_inv;      /* (nu,nu) row-major                                         */
  const void* u_min;          /* (nu) (+-inf when unbounded, mppi.py:124-126)              */
  const void* u_max;          /* (nu)                                                      */
  const void* model_params;   /* model blob, see models.hpp                          [opt] */
  const void* z;              /* TNK4 noise, J4*K*4 elements                         [opt] */
  const void* sampler_actions;/* (n_sampler_rows,T,nu)                               [opt] */
  const void* W;              /* KMPPI (T,S) interpolation operator                  [opt] */
  const void* theta;          /* KMPPI (S,nu) control points (after shift)           [opt] */
  const void* base_seq;       /* (T,nu) sequence the noise is added to and measured from;
                                 NULL = the (shifted) nominal U itself (MPPI/KMPPI)    [opt] */
  const void* process_noise_sd;/* (nx) std of the Gaussian disturbance a native model adds to every
                                 post-dynamics state -- what makes its dynamics stochastic, each of the
                                 M rollouts drawing its own (engine Philox stream, key = seed ^ tag);
                                 NULL = deterministic dynamics                         [opt] */
  /* ---- outputs (device) ---- */
  void* cost_total;           /* (K)                                                       */
  void* omega;                /* (K) normalised weights (mppi.py:258)                [opt] */
  void* cost_total_non_zero;  /* (K) exp(-(c-beta)/lambda) (mppi.py:256)             [opt] */
  void* U_out;                /* (T,nu) updated sequence (MPPI) / (S,nu) theta (KMPPI)     */
  void* action_out;           /* (u_per_command,nu)                                  [opt] */
  void* perturbed_action;     /* (K,T,nu) row-major, only written by mppi_prepare    [opt] */
  void* noise;                /* (K,T,nu) row-major, only written by mppi_prepare    [opt] */
  void* pert_cost;            /* (K) action perturbation cost (mppi.py:415), plus the
                                 smoothness cost when smooth_weight != 0               [opt] */
  void* states;               /* (K,T,nx) visited states (mppi.py:321)               [opt] */
  void* record;               /* (2 + J) shard record {beta, eta, P[J]} for the exchange   */
  /* ---- scratch ---- */
  void* workspace;            /* >= mppi_workspace_elems() elements of dtype               */
  int64_t workspace_elems;
  void* onchip_spill;         /* the on-chip form of mppi_command (noise_src = PHILOX, z = NULL), ABI 20: where the bounded noise
                                 that fits neither registers nor LDS waits for its sample's weight -- stored once behind the
                                 rollout, fetched once in the weighting phase -- instead of being GENERATED a second time;
                                 mppi_onchip_spill_elems() elements, contents meaningless between commands.  NULL: generate twice
                                 (the ABI 18 behaviour)                                                              [opt] */
  int64_t onchip_spill_elems;
  /* ---- ABI 21: the NEXT command's draw, generated inside THIS command's K3 launch ---- */
  void* next_z;               /* rng = "torch": a second TNK4 array (same pitch as z, not z itself) that mppi_weights_partial /
                                 mppi_command fill with the values torch.randn(K, T, nu) -- (K, S, nu) on a KMPPI theta problem:
                                 the shape of the rows this K3 reads -- WILL produce from generator state (next_seed,
                                 next_philox_offset), while the same launch streams this command's rows: the generator is
                                 VALU-bound, K3 is HBM-bound, four more waves per workgroup run one beside the other (reference:
                                 mppi.py:203, :378 -- the draw of command n+1 depends on nothing command n computes).  Taken only
                                 where the streaming diagonal K3 on fp32 TNK4 rows runs (one environment, (T nu) % 4 == 0);
                                 anywhere else the field is ignored.  mppi_last_next_draw() says whether it was taken; the CALLER
                                 decides at the next command whether the generator is where this assumed (mppi_noise_fill_torch
                                 otherwise) and advances it.  NULL: nothing of the kind                                 [opt] */
  uint64_t next_seed;
  uint64_t next_philox_offset;
  int32_t next_grid_blocks;   /* ATen's launch grid for that call (see mppi_noise_fill_torch) */
  int32_t next_kind;          /* MPPI_NEXT_DRAW_TORCH: as above.  MPPI_NEXT_DRAW_PHILOX (rng = the engine's generator, rows in memory):
                                 next_z receives what mppi_noise_fill_philox writes for command next_philox_offset (the caller passes
                                 p->call + 1) with key next_seed -- the rows of the NEXT command, for this problem's samples; the
                                 generator launch of small and mid-size commands disappears into K3's launch (uncoloured rows only) */
  int32_t philox_rounds;      /* rounds of the ENGINE's generator (noise_src = PHILOX, mppi_noise_fill_philox*, process noise): 0 or 10 =
                                 Philox4x32-10 (every library's default); 7 = Philox4x32-7, the fewest rounds that pass BigCrush
                                 (Salmon et al. 2011; Random123's philox4x32_R<7>, pinned by its known-answer vectors) -- a different
                                 stream, 30 % fewer multiplies in the kernel whose time they are (on-chip K1 71 -> 66 us at C3).
                                 Any other value: MPPI_E_BADARG */
  int32_t model_params_elems; /* ABI 22: elements behind `model_params`.  The built-in models with a parameter blob check it against
                                 their layout (csrc/models.hpp) and refuse a short blob with MPPI_E_BADARG instead of reading past it:
                                   MPPI_MODEL_LINEAR_GOAL  B (nx,nu) row-major | goal (nx)                          = nx*nu + nx
                                   MPPI_MODEL_MLP          W1 (H,nx+nu) | b1 (H) | W2 (nx,H) | b2 (nx) | s (1) | qx (nx) | qu (nu)
                                                           = H*(nx+nu) + H + nx*H + 2*nx + 1 + nu   (qx | qu: the diagonal quadratic
                                                           cost sum qx x^2 + sum qu u^2, appended in ABI 21; ones | zeros = sum x^2)
                                 run-time registered models (MPPI_MODEL_CUSTOM_BASE + slot) know their own blob: not checked */
} MppiProblem;

int mppi_abi_version(void);
/* sizeof(MppiProblem) as the library was compiled: bindings check their mirror against it */
int64_t mppi_problem_size(void);
const char* mppi_last_error(void);

/* rows of 4 in the TNK4 layout for a (T,nu) sequence: ceil(T*nu/4), padded so that whole
 * super-steps of lcm(4,nu) elements can be read without a tail test. */
int64_t mppi_noise_rows4(int32_t T, int32_t nu);

/* recommended row pitch (samples) of a TNK4 array for K samples: K, or K + 1 MiB worth of samples
 * when a dense row would be a multiple of 2 MiB (HBM bank aliasing between the rows K1 streams) */
int64_t mppi_noise_pitch(int32_t K, int32_t dtype);

/* elements of dtype the workspace must hold for this problem */
int64_t mppi_workspace_elems(const MppiProblem* p);

/* elements of dtype p->onchip_spill needs for the on-chip command to generate nothing twice (0: fp64, or everything fits on chip);
 * at C3 (K = 65536, T = 64, nu = 12): 87 of the sample's 192 rows-of-4 wait there (105 stay in registers and LDS), in whole
 * weighting tiles: 90 rows x K x 16 B = 94 MB */
int64_t mppi_onchip_spill_elems(const MppiProblem* p);

/* 1 if a fused rollout kernel exists for (model_id, nx, nu, dtype, hidden), else 0 */
int mppi_model_supported(int32_t model_id, int32_t nx, int32_t nu, int32_t dtype, int32_t hidden);

/* replaces torch.randn(K,T,nu) at mppi.py:203 (called from :378 / KMPPI :660): fills
 * p->z (written, despite the const) in TNK4 layout with the Philox stream the fused mode uses */
int mppi_noise_fill_philox(const MppiProblem* p, void* z_tnk4, void* stream);

/* the same stream with the colouring of mppi.py:201-206 applied by the generator: writes
 * eps = chol(Sigma) z + mu (full Sigma) in TNK4 layout; run the rest of the command with
 * p->noise_coloured = 1 and K1 / K3 take their diagonal-Sigma form (no per-sample L z in the
 * HBM-bound kernels).  MPPI_E_UNSUPPORTED for control widths without a compiled instantiation. */
int mppi_noise_fill_philox_coloured(const MppiProblem* p, void* eps_tnk4, void* stream);

/* rng = "torch" (the reference's own draw, mppi.py:203 `torch.randn(K, T, nu)` on the controller's device, fp32): the VALUES
 * that call would produce from generator state (seed, philox_offset), written straight into a TNK4 array of row pitch
 * `pitch` (>= K, in rows-of-4) -- bit for bit what ATen's normal_ kernel computes (Philox4x32-10 keyed `seed`, subsequence
 * = thread index, rocrand's Box-Muller; thread idx of 256 * grid_blocks threads gives element idx + 256 grid_blocks (4 m +
 * i) component i of its m-th block), enumerated by destination.  `grid_blocks` is ATen's launch grid for that call:
 * min(multiProcessorCount * (maxThreadsPerMultiProcessor / 256), ceil(K T nu / 256)); the CALLER advances the generator by
 * ((K T nu - 1) / (1024 grid_blocks) + 1) * 4 like ATen does, so that every later draw of the process is unchanged.
 * Needs (T nu) % 4 == 0; MPPI_E_UNSUPPORTED otherwise (draw with torch.randn and use MPPI_NOISE_KTN / mppi_noise_from_ktn). */
int mppi_noise_fill_torch(void* z_tnk4, int64_t K, int32_t T, int32_t nu, int64_t pitch, uint64_t seed, uint64_t philox_offset,
                          int32_t grid_blocks, void* stream);

/* Test / debug seam: the process-noise normals the fused multi-rollout K1 (rollout_samples M in 2..4, p->process_noise_sd
 * set) draws in-kernel for command p->call -- its own Philox key (seed ^ tag), counter (sample, (t*4 + m)*ceil(nx/4) +
 * block, call) -- written to `out` as (M,K,T,nx) row-major.  Generated by the same device function the kernel calls, so a
 * checker (the oracle) can be fed exactly the numbers the rollout consumed. */
int mppi_process_noise_export(const MppiProblem* p, void* out_mktx, void* stream);

/* (K,T,nu) row-major standard normals (the reference's layout, mppi.py:203) -> TNK4 */
int mppi_noise_from_ktn(const MppiProblem* p, const void* z_ktn, void* z_tnk4, void* stream);

/* KMPPI._compute_perturbed_action_and_noise (mppi.py:657-666): eps_S = colour(z_S) over the S
 * support points (p->z / Philox, J = S*nu), ctrl = clamp(theta + eps_S), raw action
 * v[t] = sum_s W[t,s] ctrl[s]  (the constant-operator form of the vmap'd solve at :630-655),
 * written in TNK4 layout (J = T*nu) to v_tnk4. */
int mppi_kmppi_interp(const MppiProblem* p, void* v_tnk4, void* stream);

/* KMPPI's bookkeeping on the nominal sequences, one small launch each (a host-side formulation costs a roll, a
 * copy and two GEMM launches per command):
 *   mppi_kmppi_shift       KMPPI.shift_nominal_trajectory (mppi.py:232-238, :617-619):
 *                          U_out (T,nu) = roll(U, -1) with u_init (nu) in the last row;  theta_out (S,nu) = W_shift (S,S) theta
 *   mppi_kmppi_trajectory  deparameterize_to_trajectory_single (mppi.py:682, :647-648):  U_out (T,nu) = W (T,S) theta (S,nu)
 * Outputs must not alias inputs. */
int mppi_kmppi_shift(int32_t dtype, int32_t T, int32_t S, int32_t nu, const void* U, const void* u_init,
                     const void* theta, const void* W_shift, void* U_out, void* theta_out, void* stream);
int mppi_kmppi_trajectory(int32_t dtype, int32_t T, int32_t S, int32_t nu, const void* W, const void* theta,
                          void* U_out, void* stream);
/* ABI 22: behind a KMPPI update, ONE launch for the trajectory of the new control points and for both sequences as the next
 * command's shift will want them: U_out = W theta (mppi.py:682); theta_shift_out = W_shift theta (:617-619); U_shift_out =
 * roll(U_out, -1) with u_init in the last row (:232-238) -- the bits mppi_kmppi_trajectory followed by mppi_kmppi_shift produce.
 * A loop that shifts on every command (mppi.py:240's default) then launches nothing for the shift. */
int mppi_kmppi_after_update(int32_t dtype, int32_t T, int32_t S, int32_t nu, const void* W, const void* W_shift, const void* theta,
                            const void* u_init, void* U_out, void* theta_shift_out, void* U_shift_out, void* stream);

/* Host -> device hand-over of a SMALL buffer (the state of a closed loop: mppi.py:262-264) inside the launch packet of
 * a one-wave kernel: 4..2048 bytes, a multiple of 4, from ordinary (pageable) host memory, which is consumed before
 * the call returns; ordered on `stream` like any kernel.  ~5 us of host time against ~20 us for a pageable hipMemcpy. */
int mppi_upload_small(const void* src_host, int64_t nbytes, void* dst_device, void* stream);

/* SMPPI.shift_nominal_trajectory (mppi.py:488-492) plus the base sequence of the next command (:540), one launch:
 * U_out = roll(U, -1) with u_init in the last row; A_out = roll(action_sequence, -1) with the last row repeated;
 * B_out = A_out + U_out * dt (what MppiProblem.base_seq points at).  Outputs must not alias inputs. */
int mppi_smppi_shift(int32_t dtype, int32_t T, int32_t nu, const void* U, const void* u_init, const void* A, double dt,
                     void* U_out, void* A_out, void* B_out, void* stream);

/* K1 -- replaces _compute_total_cost_batch (mppi.py:407-417) = _sample_noise colouring
 * (:201-206), _compute_perturbed_action_and_noise (:375-385), _sample_specific_actions
 * (:387-400), _bound_action (:419-420), _compute_action_cost (:186-199) and
 * _compute_rollout_costs_single (:297-332) for a native model: writes cost_total (K) and the
 * per-block minima into the workspace. */
int mppi_rollout_cost(const MppiProblem* p, void* stream);

/* K1 for KMPPI with the interpolation INSIDE the launch (mppi.py:653-670 + :407-417): `p` as for
 * mppi_kmppi_interp -- the trajectory problem (T, U, state, cost_total, sampler rows ...) whose noise_src / z /
 * noise_pitch / seed / call describe the SUPPORT-point stream (S rows of nu; MPPI_NOISE_TNK4 or
 * MPPI_NOISE_PHILOX), plus S, theta (S,nu), W (T,S).  Bounded control points stay in the lane's registers,
 * v[t] = sum_s W[t,s] theta'[s] is formed four timesteps at a time on the matrix cores and consumed by the
 * rollout at once: nothing of shape (K,T,nu) is written or read.  Same cost_total as mppi_kmppi_interp followed by
 * mppi_rollout_cost(noise_src = MPPI_NOISE_ACTIONS) up to fp32 rounding of the sum over s.
 * fp32, diagonal Sigma, nu in {4,8,12,16}, S*nu <= 384, native / JIT models on the per-lane path; anything else
 * returns MPPI_E_UNSUPPORTED before any launch -> run the two calls instead. */
int mppi_rollout_cost_kmppi(const MppiProblem* p, void* stream);
/* process-wide count of mppi_rollout_cost_kmppi calls that launched (tests, bench) */
int64_t mppi_stat_kmppi_fused_rollouts(void);
/* ABI 19 -- one KMPPI command (mppi.py:672-688) from one call: K1 with the interpolation inside (`p`, as above) and the
 * control-point update theta += sum_k omega_k noise_theta_k (mppi.py:679-681) on `theta_problem` -- the (K, S, nu) problem whose
 * "nominal sequence" U is theta, U_out the new theta, `record` / cost_total (== p's) / lambda as for mppi_weights_partial +
 * mppi_finalize, on THE SAME workspace as `p`.  Where the kernel can (LDS room for the column sums; no sampler rows) it reduces
 * its workgroups' part of the update from the bounded control points the lanes STILL HOLD -- weights relative to the
 * workgroup's own minimum, one partial record per 256 samples, the algebra of the on-chip mppi_command -- and a small launch
 * combines the records: the stand-alone K3 that re-creates every sample's S*nu control-point rows is gone.  Elsewhere:
 * mppi_rollout_cost_kmppi + mppi_weights_partial + mppi_finalize on the two problems.  MPPI_E_UNSUPPORTED (before any launch)
 * where the fused-interpolation kernel does not exist: run the two-launch form.  omega / cost_total_non_zero of the theta
 * problem are written when given (they are functions of cost_total and the record otherwise). */
int mppi_command_kmppi(const MppiProblem* p, const MppiProblem* theta_problem, int apply, void* stream);
/* process-wide count of mppi_command_kmppi calls whose theta update was reduced inside K1 */
int64_t mppi_stat_kmppi_onchip_updates(void);

/* generic path (user callbacks stay Python callables, mppi.py:63-64): everything of
 * _compute_total_cost_batch except the rollout loop: writes perturbed_action, noise (K,T,nu)
 * and pert_cost (K).  Also serves the lazily materialised public attributes
 * `perturbed_action` / `noise` (mppi.py:383-385) of the fused path. */
int mppi_prepare(const MppiProblem* p, void* stream);

/* generic path: block minima of a cost_total (K) that Python assembled (mppi.py:416) */
int mppi_cost_block_min(const MppiProblem* p, void* stream);

/* K3 -- replaces _compute_weighting (mppi.py:254-259, :12-13) and the weighted sum of
 * einsum('k,ktn->tn', omega, noise) (:268; KMPPI :679 on noise_theta): per-block partial
 * eta and P[j] relative to this shard's own beta, into the workspace. */
int mppi_weights_partial(const MppiProblem* p, void* stream);

/* K4 -- fixed-order reduction of the block partials into p->record = {beta, eta, P[J]};
 * with `apply` != 0 also finishes the single-shard command: U_out = shift(U) + P/eta
 * (mppi.py:270; KMPPI: theta_out, then U = W theta is applied by the host), action_out
 * (:271-275), omega / cost_total_non_zero (:256-258). */
int mppi_finalize(const MppiProblem* p, int apply, void* stream);

/* K1 + K3 + K4 in one call (the fused path of one MPPI._command, mppi.py:261-275): exactly
 * mppi_rollout_cost, then mppi_weights_partial on the rows K1 read or generated, then
 * mppi_finalize(apply).  Nothing is launched when K1 refuses the problem (negative status).
 * Small problems (K <= 16384, T*nu <= 256, diagonal Sigma or a coloured stream, no sampler rows, rows
 * in memory: p->z set) are run as ONE launch -- every workgroup appends its part of K3 to its rollout
 * and the last one to finish combines the partial records and applies K4 -- IF the caller passes
 * omega == NULL and cost_total_non_zero == NULL; both are functions of the outputs that ARE written:
 *   cost_total_non_zero = exp(-(cost_total - record[0]) / lambda),  omega = cost_total_non_zero / record[1].
 * The WORKSPACE MUST BE ZERO-FILLED ONCE before its first use: the arrival ticket of this form lives in the LAST 4
 * elements of the buffer as the caller describes it -- at workspace + workspace_elems - 4, a place that does not
 * depend on the problem's shape -- and every command leaves it at zero.  Problems of different shapes may therefore
 * share one buffer, as long as every one of them is handed the SAME (workspace, workspace_elems) pair (sized for the
 * largest: mppi_workspace_elems) and they run in stream order.  A launch that faulted may leave the ticket non-zero:
 * zero those 4 elements again (hipMemsetAsync) before reusing the buffer. */
int mppi_command(const MppiProblem* p, int apply, void* stream);
/* process-wide count of mppi_command calls that ran in the single-launch form (tests, bench) */
int64_t mppi_stat_single_launch_commands(void);
/* The ON-CHIP form of mppi_command (ABI 18, csrc/rollout_onchip.hpp): noise_src == MPPI_NOISE_PHILOX with p->z == NULL
 * ("no row array: the normals are a pure function of seed, call, sample, row") on a fused fp32 model with a diagonal or full
 * Sigma, MPPI or SMPPI (base_seq / noise_rescale / smooth_weight honoured; no S), rollout_samples <= 1, no sampler rows, no `states`, one
 * environment.  ONE launch generates each sample's normals, rolls out, keeps the bounded noise on chip (accumulation
 * registers + LDS; what does not fit is generated a second time) and reduces it into one partial record
 * {beta_b, eta_b, P_b} per 256-sample workgroup, relative to the workgroup's own minimum; a second, small launch
 * combines the records in workgroup order (the algebra of mppi_combine) and applies K4.  No (K,T,nu) array is
 * written or read.  Outputs as for the other forms (cost_total, record, U_out, action_out; omega / cost_total_non_zero
 * when given).  Problems outside that scope with p->z == NULL run K1 + K3 with the rows generated twice, as before.
 * MPPI_ONCHIP=0 in the environment disables the form (A/B runs).  Count of commands that took it: */
int64_t mppi_stat_onchip_commands(void);
/* Round 6 (still ABI 22: additive): on-chip commands whose K1 ran with TWO waves per 64-sample group (csrc/rollout_onchip_pair.hpp:
 * the waves of a pair own alternating chunks of the horizon, each generates, keeps and later sums its own rows, the rollout's state
 * goes from one to the other through LDS -- a second wave per SIMD hides the first one's waits; bit for bit the results of the
 * one-wave kernel).  Taken by the plain command (no |noise| cost, u_scale 1, no SMPPI terms, diagonal Sigma) when the caller's
 * `onchip_spill` array has mppi_onchip_spill_elems() elements (the larger of the two forms' needs); MPPI_ONCHIP_PAIR=0 / 1 in the
 * environment forces the one-wave / allows the two-wave kernel (read at every launch). */
int64_t mppi_stat_onchip_pair_launches(void);
/* ABI 22: launches of the split-operand matrix-core K1 of MPPI_MODEL_MLP (csrc/rollout_mlp_split.hip) in this process -- which
 * kernel a dense-MLP command ran is otherwise invisible to the caller (the per-lane form gives the same results to parity) */
int64_t mppi_stat_mlp_split_launches(void);
/* ABI 19: the form the CALLING THREAD's last mppi_command took (thread-local; the two counters above are process-wide and
 * cannot answer "did my command run on chip" once two controllers command from two threads). */
#define MPPI_FORM_NONE 0           /* no command yet on this thread, or the last one failed before its first launch */
#define MPPI_FORM_STREAMING 1      /* K1 + K3 + K4 */
#define MPPI_FORM_SINGLE_LAUNCH 2  /* small problems: one launch */
#define MPPI_FORM_ONCHIP 3         /* on-chip K1 + finalize_blocks */
int mppi_last_command_form(void);
/* ABI 21: 1 when the calling thread's last mppi_weights_partial / mppi_command / mppi_command_sharded / mppi_command_kmppi
 * also generated p->next_z (see MppiProblem.next_z), else 0 */
int mppi_last_next_draw(void);

/* K5 -- multi-GPU: combine `n_shards` records (all-gathered, rank order) exactly the same way
 * on every rank: beta = min beta_g; s_g = exp(-(beta_g-beta)/lambda); eta = sum s_g eta_g;
 * U_out = shift(U) + sum s_g P_g / eta; rescales this shard's omega. */
int mppi_combine(const MppiProblem* p, const void* records, int32_t n_shards, void* stream);
/* ABI 22: the same combine with every shard's record read WHERE ITS K4 LEFT IT (record_ptrs[g] -> the (2 + J) elements of shard g:
 * memory of this device, or of a peer this device can access) instead of from one gathered array: no copy, no collective -- the
 * staged exchange of a device group (mppi_group_*).  The caller orders the launch behind the writers (events).  Same arithmetic,
 * same order, same bits as mppi_combine on the gathered records.  n_shards <= MPPI_MAX_GROUP. */
#define MPPI_MAX_GROUP 16
int mppi_combine_ptrs(const MppiProblem* p, const void* const* record_ptrs, int32_t n_shards, void* stream);

/* Multi-GPU exchange (the reference has none): K is sharded over one process per GPU,
 * the only data-path collective of a command is ONE all-gather of the (2 + T*nu)-element shard
 * record, issued through RCCL's C API on the CALLER'S stream (RCCL is bound at run time with
 * dlsym/dlopen; mppi_dist_available() says whether it was found).  Bootstrap: rank 0 obtains an id
 * with mppi_dist_unique_id (128 bytes), ships it to the other ranks by any means (e.g. a
 * torch.distributed broadcast at start-up), every rank calls mppi_dist_init.
 *   mppi_exchange_combine  ncclAllGather(p->record -> records[world_size][2+J]) then mppi_combine (K5)
 *   mppi_command_sharded   mppi_command(apply = 0) + mppi_exchange_combine: one sharded command of
 *                          the fused path as ONE call, five launches on one stream, no host hop    */
int mppi_dist_available(void);
int mppi_dist_unique_id(void* id128);
int mppi_dist_init(const void* id128, int32_t rank, int32_t world_size, void** comm_out);
int mppi_dist_destroy(void* comm);
int mppi_exchange_combine(const MppiProblem* p, void* comm, void* records, int32_t world_size, void* stream);
int mppi_command_sharded(const MppiProblem* p, void* comm, void* records, int32_t world_size, void* stream);

/* One process, N devices: communicators of all listed devices from ncclCommInitAll (comms_out[ndev]; each is
 * destroyed with mppi_dist_destroy), and the exchange of one command on all of them from one call: the N all-gathers of the shard
 * records (problems[g]->record -> records[g], (ndev, 2 + T nu) elements) inside ONE ncclGroupStart / ncclGroupEnd, each on
 * streams[g] -- the stream device devs[g]'s K1 / K3 / K4 were issued on --, then mppi_combine on every device.  The calling
 * thread's current device is left as it was.  MPPI_E_UNSUPPORTED: no RCCL, or a device listed twice (RCCL takes one rank per
 * device: a caller that shards over ONE device for testing stages the records itself and calls mppi_combine). */
int mppi_dist_init_all(int32_t ndev, const int32_t* devs, void** comms_out);
int mppi_exchange_combine_all(int32_t ndev, const int32_t* devs, const MppiProblem* const* problems, void* const* comms,
                              void* const* records, void* const* streams);

/* ABI 22 -- one process, N devices, the WHOLE command from one place (csrc/group.hip; reference caller: mppi.py:876-898, one
 * process stepping one environment).  A group owns one worker thread per listed device (that device current in it, once): the
 * caller fills the N problem blocks and hands each over with mppi_group_submit -- the worker starts issuing that device's K1 / K3 /
 * K4 (mppi_command(apply = 0), or mppi_command_kmppi when a theta problem comes with it) at once, while the caller fills the next
 * block -- and mppi_group_wait commits the command (all N parts are in), after which every worker, behind a host barrier of the
 * workers (nobody enters a collective that a failed peer would never join), issues the exchange of the (2 + T nu)-element shard
 * records into records[g] ((ndev, 2 + T nu) elements on device g) and mppi_combine on its stream, and returns once all have ISSUED
 * their part (no device synchronisation).  The exchange: `comms` from mppi_dist_init_all -> ncclAllGather on each device's own
 * communicator (one thread per device: no ncclGroupStart / End); comms = NULL ("staged": a device listed twice -- the one-GPU test
 * rig -- or no RCCL) -> an event behind K4, and K5 reads the other shards' records in place behind their events
 * (mppi_combine_ptrs) where every device can access every other's memory, else after peer copies into records[g]; ndev <=
 * MPPI_MAX_GROUP.  The blocks are COPIED at submit; the device buffers they point to must stay valid until the stream has run.
 *   mppi_group_broadcast   optional, before the first submit of a command: `nbytes` at `src` (memory of devs[0], ready in the
 *                          order of stream0) are copied to dst[g] in front of device g's K1 (dst[g] == src or NULL: nothing)
 *   mppi_group_wait        forms[g] / next_draws[g]: what mppi_last_command_form() / mppi_last_next_draw() read on worker g.
 *                          Returns the first device's own error (message: "device d: ..."); the others report
 *                          MPPI_E_GROUP_PEER and issue nothing further.  Called with parts missing it abandons the command
 *   mppi_group_abort       abandon the command being assembled (the caller could not fill every block): the devices that did
 *                          get their part have issued K1 / K3 / K4 -- harmless, nothing was applied -- and skip the exchange
 * One caller thread per group (the reference's controller is not re-entrant either). */
int mppi_group_create(int32_t ndev, const int32_t* devs, void* const* comms, void** group_out);
int mppi_group_destroy(void* group);
int mppi_group_size(void* group);
int mppi_group_broadcast(void* group, const void* src, int64_t nbytes, void* const* dst, void* stream0);
int mppi_group_submit(void* group, int32_t index, const MppiProblem* p, const MppiProblem* theta_problem, void* records, void* stream);
int mppi_group_wait(void* group, int32_t* forms, int32_t* next_draws);
int mppi_group_abort(void* group);

/* User models.  The reference's plugin API is "any Python callable" (mppi.py:63-64); the fused
 * equivalent is a device functor {step, cost, terminal} that pytorch_mppi_amd/jit.py wraps around
 * the user's C++ snippets, compiles with hipcc against csrc/rollout.hpp into its own shared object
 * and registers here.  `rollout_f32` / `rollout_f64`: int (*)(const void* kargs, void* stream), the
 * object's own instantiation of K1 (kargs = the engine-internal typed argument block that
 * mppi_rollout_cost builds).  model_id = MPPI_MODEL_CUSTOM_BASE + slot, slot in [0, 64). */
int mppi_register_model(int32_t model_id, int32_t nx, int32_t nu, void* rollout_f32, void* rollout_f64);

/* Measurement hooks (bench.py).  mppi_profile_enable(every): while enabled (every >= 1), each K1
 * launch (mppi_rollout_cost) stamps wall_clock64() at workgroup entry and exit -- min(entry) /
 * max(exit) per launch is the kernel's own span on the device clock: no extra packets, nothing measurable
 * added, but it misses the dispatch ramp in front of the first wave and the drain behind the last (a
 * constant ~0.8-1.1 us against `rocprofv3 --kernel-trace` on the same launches) -- and every `every`-th
 * launch is additionally made with hipExtLaunchKernelGGL start/stop events.  Those are NOT neutral: the
 * start event is a marker packet in front of the kernel and an event-carrying dispatch ends with a
 * system-scope release (~3 us longer than the same kernel launched plainly), so pass a large `every`
 * (e.g. 1 << 30) inside a timed region and sample events in a pass of their own.
 * mppi_profile_enable(0) switches both off.
 * mppi_profile_read_launches synchronises and returns, per launch since the last read (at most `capacity`;
 * `count` = how many there were), the device-clock span and the event-pair time in MICROSECONDS (-1 where
 * a launch carried no events), and clears the record; mppi_profile_read2 returns the sums in milliseconds
 * with the number of launches stamped (`count`) and of launches with events (`count_events`);
 * mppi_profile_read the event part only. */
int mppi_profile_enable(int every);
int mppi_profile_read(double* sum_ms, int64_t* count_events);
int mppi_profile_read2(double* sum_ms_events, double* sum_ms_device, int64_t* count, int64_t* count_events);
int mppi_profile_read_launches(double* device_us, double* dispatch_us, int64_t capacity, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_AMD_H */
