// capi.hip -- the extern "C" boundary declared in include/mppi_amd.h.
// Validates the problem block, carves the caller's workspace, builds the typed kernel argument
// block and dispatches.  No allocation, no device synchronisation, no global state besides a
// thread-local error string.
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <vector>
#include <cstring>
#include <hip/hip_ext.h>
#include "dispatch.hpp"
#include "update.hpp"

using namespace mppi;

static thread_local char g_err[256] = "";

static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
// message hook for the other translation units of the C-ABI (dist.hip)
int mppi_fail_message(int code, const char* msg) { return fail(code, msg); }
static int hipfail(int code, const char* where) {
  if (code > 0)
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString((hipError_t)code));
  else if (code == MPPI_E_UNSUPPORTED)
    snprintf(g_err, sizeof(g_err), "%s: no kernel instantiated for this model/nx/nu/dtype", where);
  else if (code < 0)
    snprintf(g_err, sizeof(g_err), "%s: bad argument", where);
  return code;
}

static int gcd4(int nu) { return nu % 4 == 0 ? 4 : (nu % 2 == 0 ? 2 : 1); }

extern "C" int64_t mppi_noise_rows4(int32_t T, int32_t nu) {
  if (T <= 0 || nu <= 0) return 0;
  const int g = gcd4(nu), p4 = nu / g, tt = 4 / g;
  return (int64_t)((T + tt - 1) / tt) * p4;
}

// Row pitch (in samples) the engine recommends for a TNK4 array of K samples.  A row is K*4 elements;
// when that is a multiple of 2 MiB, consecutive rows land on the same HBM banks and the row-streaming
// K1 (9 rows in flight per lane, all workgroups at the same offset of their rows) loses a quarter of
// its bandwidth: measured 53 % of 8 TB/s at K = 262144 (4 MiB rows) against 71 % at K = 327680
// (5 MiB) -- profiles/r02_k1_row_stride.txt.  Such rows are padded by 1 MiB.
extern "C" int64_t mppi_noise_pitch(int32_t K, int32_t dtype) {
  if (K <= 0) return 0;
  const int64_t bps = dtype == MPPI_F64 ? 32 : 16;               // bytes of one row-of-4 per sample
  const int64_t row = (int64_t)K * bps;
  if (row >= (2 << 20) && row % (2 << 20) == 0) return K + (1 << 20) / bps;
  return K;
}

// ---- measurement hook -------------------------------------------------------------------------
// Two clocks per K1 launch while enabled:
//  * the kernel's own span on the device wall clock (min workgroup entry .. max exit), EVERY launch: no
//    extra packets, no perturbation.  It misses the dispatch ramp in front of the first wave and the
//    end-of-kernel drain behind the last one: rocprofv3 --kernel-trace reads a constant ~0.8-1.1 us more
//    on the same launches (profiles/r03_k1_clock_calibration.txt);
//  * on every `every`-th launch a hipExtLaunchKernelGGL start/stop event pair.  NOT a neutral clock
//    (tools/micro/event_clock.hip, profiles/r03_event_clock.txt): the start event is a marker packet in
//    front of the kernel, an event-carrying dispatch ends with a system-scope release, and
//    hipEventElapsedTime of kernel-bound events is end - end -- an event-attached launch reads ~3 us
//    longer than the same kernel launched plainly.  Kept because HIP events are the conventional clock;
//    bench.py takes it in a separate pass behind the timed region.
namespace {
constexpr int PROF_MAX = 2048;             // launches per mppi_profile_enable .. mppi_profile_read_launches window
int g_prof_every = 0;                      // 0 = off; N = HIP events on every N-th K1 launch
int g_prof_n = 0;                          // launches stamped since the last read
hipEvent_t g_prof_ev[PROF_MAX][2];         // event pairs, created on first use
bool g_prof_has_ev[PROF_MAX];              // does launch i carry events?
int g_prof_created = 0;
unsigned long long* g_prof_ts = nullptr;   // device: PROF_MAX launches x STAMP_SLOTS x {min entry, max exit} (common.hpp stamp_entry)
constexpr size_t PROF_TS_WORDS = (size_t)2 * mppi::STAMP_SLOTS * PROF_MAX;
double prof_dispatch_ms(int i) {           // < 0: launch i carried no events
  if (!g_prof_has_ev[i]) return -1.0;
  float ms = 0;
  if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]) != hipSuccess) return -1.0;
  return (double)ms;
}
}  // namespace
namespace mppi {
bool profile_next_events(hipEvent_t* start, hipEvent_t* stop, unsigned long long** tstamp) {
  if (tstamp) *tstamp = nullptr;
  *start = *stop = nullptr;
  if (g_prof_every <= 0 || g_prof_n >= PROF_MAX) return false;
  const int i = g_prof_n++;
  if (tstamp && g_prof_ts) *tstamp = g_prof_ts + (size_t)2 * STAMP_SLOTS * i;
  g_prof_has_ev[i] = false;
  if (g_prof_every >= (1 << 30) || i % g_prof_every != 0) return false;     // (1 << 30: stamps only, no launch carries events)
  if (i >= g_prof_created) {
    // pairs are created densely up to i so that index == launch number
    for (int j = g_prof_created; j <= i; ++j) {
      if (hipEventCreate(&g_prof_ev[j][0]) != hipSuccess) return false;
      if (hipEventCreate(&g_prof_ev[j][1]) != hipSuccess) return false;
      g_prof_created = j + 1;
    }
  }
  *start = g_prof_ev[i][0];
  *stop = g_prof_ev[i][1];
  g_prof_has_ev[i] = true;
  return true;
}
}  // namespace mppi
extern "C" int mppi_profile_enable(int every) {
  g_prof_every = every > 0 ? every : 0;
  if (every > 0) {
    g_prof_n = 0;
    // measurement set-up (outside any timed region): stamp slots {min = ~0, max = 0}
    if (!g_prof_ts && hipMalloc((void**)&g_prof_ts, sizeof(unsigned long long) * PROF_TS_WORDS) != hipSuccess) g_prof_ts = nullptr;
    // arm the window: every slot {0, 0} (stamp_entry / stamp_exit are both atomicMax)
    if (g_prof_ts && hipMemset(g_prof_ts, 0, sizeof(unsigned long long) * PROF_TS_WORDS) != hipSuccess) {
      (void)hipFree(g_prof_ts);
      g_prof_ts = nullptr;
    }
  }
  return 0;
}
extern "C" int mppi_profile_read_launches(double* device_us, double* dispatch_us, int64_t capacity, int64_t* count) {
  const int n = g_prof_n;
  if (count) *count = n;
  static std::vector<unsigned long long> slots, host;
  host.resize((size_t)2 * PROF_MAX);
  bool have_dev = false;
  if (g_prof_ts && n > 0) {
    slots.resize(PROF_TS_WORDS);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(slots.data(), g_prof_ts, sizeof(unsigned long long) * 2 * mppi::STAMP_SLOTS * n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hipfail((int)e, "mppi_profile_read_launches");
    for (int i = 0; i < n; ++i) {              // the launch's span: earliest entry .. latest exit over its workgroups' slots
      unsigned long long nlo = 0ull, hi = 0ull;     // slot[0] = max of ~entry over the slot's workgroups, slot[1] = max of exit
      for (int q = 0; q < mppi::STAMP_SLOTS; ++q) {
        const unsigned long long a = slots[((size_t)i * mppi::STAMP_SLOTS + q) * 2], b = slots[((size_t)i * mppi::STAMP_SLOTS + q) * 2 + 1];
        nlo = a > nlo ? a : nlo;
        hi = b > hi ? b : hi;
      }
      host[2 * i] = nlo != 0ull ? ~nlo : ~0ull;     // (no stamp at all: an empty span)
      host[2 * i + 1] = hi;
    }
    have_dev = true;
  }
  int dev_id = 0, khz = 100000;
  (void)hipGetDevice(&dev_id);
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev_id);
  for (int i = 0; i < n && i < capacity; ++i) {
    if (device_us)
      device_us[i] = (have_dev && host[2 * i + 1] > host[2 * i]) ? (double)(host[2 * i + 1] - host[2 * i]) / (double)khz * 1e3 : -1.0;
    if (dispatch_us) {
      const double ms = prof_dispatch_ms(i);
      dispatch_us[i] = ms < 0 ? -1.0 : ms * 1e3;
    }
  }
  g_prof_n = 0;
  return 0;
}
extern "C" int mppi_profile_read2(double* sum_ms_events, double* sum_ms_device, int64_t* count, int64_t* count_events) {
  static double dev_us[PROF_MAX], disp_us[PROF_MAX];
  int64_t n = 0;
  if (int e = mppi_profile_read_launches(dev_us, disp_us, PROF_MAX, &n)) return e;
  double dev = 0, ev = 0;
  int64_t ne = 0;
  for (int i = 0; i < n; ++i) {
    if (dev_us[i] > 0) dev += dev_us[i] * 1e-3;
    if (disp_us[i] >= 0) { ev += disp_us[i] * 1e-3; ++ne; }
  }
  if (sum_ms_device) *sum_ms_device = dev;
  if (sum_ms_events) *sum_ms_events = ev;
  if (count) *count = n;
  if (count_events) *count_events = ne;
  return 0;
}
extern "C" int mppi_profile_read(double* sum_ms, int64_t* count_events) {
  return mppi_profile_read2(sum_ms, nullptr, nullptr, count_events);
}

namespace {
// run-time registered (JIT-compiled) models: see mppi_register_model
typedef int (*custom_rollout_fn)(const void* kargs, void* stream);
struct CustomModel { int nx, nu; custom_rollout_fn f32, f64; };
constexpr int MAX_CUSTOM = 64;
CustomModel g_custom[MAX_CUSTOM] = {};

struct Carve { int nb1, nkc, Jpad, R; int64_t total; };

Carve carve(const MppiProblem* p) {
  Carve c;
  const int64_t J4 = mppi_noise_rows4(p->T, p->nu);
  c.Jpad = (int)(J4 * 4);
  if (c.Jpad % UPD_TJ) c.Jpad += UPD_TJ - c.Jpad % UPD_TJ;
  c.nb1 = (p->K + WAVE - 1) / WAVE;            // one cost minimum per wave of 64 samples
  // samples per lane in K3: enough k-chunks to fill the chip, at most 8 loads in flight per lane
  int R = 4;
  const int njt = c.Jpad / UPD_TJ;
  while (R > 1 && (int64_t)((p->K + BLOCK * R - 1) / (BLOCK * R)) * njt < 512) R >>= 1;
  if (p->noise_src == MPPI_NOISE_KTN) R = 1;   // lane-per-column K3: blocks come from the k axis (J/256 column groups only)
  // tuning knob for tools/ sweeps, read once
  static const int r_env = [] { const char* e = getenv("MPPI_K3_R"); return e ? atoi(e) : 0; }();
  if (r_env == 1 || r_env == 2 || r_env == 4 || r_env == 8) R = r_env;
  c.R = R;
  c.nkc = (p->K + BLOCK * R - 1) / (BLOCK * R);
  const int64_t ne = p->num_envs > 1 ? p->num_envs : 1;
  // sized for the finest chunking (R = 1) so one workspace serves every noise_src of the same problem
  const int64_t nkc_max = (p->K + BLOCK - 1) / BLOCK;
  c.total = ne * ((int64_t)c.nb1 + nkc_max + nkc_max * c.Jpad) + 4;     // + the arrival ticket of the single-launch command
  return c;
}

template <typename T>
int make_args(const MppiProblem* p, KArgs<T>& a) {
  if (p == nullptr) return fail(MPPI_E_BADARG, "null problem");
  if (p->K <= 0 || p->T <= 0 || p->nx <= 0 || p->nu <= 0) return fail(MPPI_E_BADARG, "bad dims");
  if (!(p->lambda_ > 0)) return fail(MPPI_E_BADARG, "lambda_ must be > 0");
  if (p->philox_rounds != 0 && p->philox_rounds != 7 && p->philox_rounds != 10) return fail(MPPI_E_BADARG, "philox_rounds must be 0, 7 or 10");
  if (!p->U || !p->u_init || !p->noise_mu || !p->noise_L || !p->sigma_inv || !p->u_min || !p->u_max)
    return fail(MPPI_E_BADARG, "missing parameter array");
  if (p->n_sampler_rows > 0 && !p->sampler_actions) return fail(MPPI_E_BADARG, "sampler rows without actions");
  {
    // ABI 22: the built-in models' parameter blobs have a stated length (a caller still passing an older, shorter layout is
    // refused instead of being read past its end)
    int64_t need = 0;
    if (p->model_id == MPPI_MODEL_MLP) need = (int64_t)p->hidden * (p->nx + p->nu) + p->hidden + (int64_t)p->nx * p->hidden + 2 * p->nx + 1 + p->nu;
    if (p->model_id == MPPI_MODEL_LINEAR_GOAL) need = (int64_t)p->nx * p->nu + p->nx;
    if (need > 0 && p->model_params != nullptr && p->model_params_elems < need) {
      char msg[200];
      snprintf(msg, sizeof(msg), "model_params holds %d elements (model_params_elems), this model's blob needs %lld (layout: include/mppi_amd.h)",
               (int)p->model_params_elems, (long long)need);
      return fail(MPPI_E_BADARG, msg);
    }
  }
  const Carve c = carve(p);
  if (!p->workspace || p->workspace_elems < c.total) return fail(MPPI_E_WORKSPACE, "workspace too small");
  a.K = p->K; a.Tn = p->T; a.nx = p->nx; a.nu = p->nu; a.J = p->T * p->nu;
  a.J4 = (int)mppi_noise_rows4(p->T, p->nu);
  a.k_offset = p->k_offset;
  a.zp = p->noise_pitch > 0 ? p->noise_pitch : p->K;
  if (a.zp < p->K) return fail(MPPI_E_BADARG, "noise_pitch < K");
  a.model_id = p->model_id; a.diag = p->sigma_diagonal; a.abs_cost = p->noise_abs_cost;
  a.null_action = p->sample_null_action; a.n_sampler = p->n_sampler_rows;
  a.state_per_sample = p->state_per_sample; a.shift = p->shift; a.use_terminal = p->use_terminal;
  a.noise_src = p->noise_src; a.u_per_command = p->u_per_command; a.hidden = p->hidden;
  a.coloured = p->noise_coloured != 0;
  a.model_flags = p->model_flags;
  a.spill = (T*)p->onchip_spill;
  a.spill_cap = p->onchip_spill != nullptr ? p->onchip_spill_elems : 0;
  a.seven = p->philox_rounds == 7;
  a.lambda_ = (T)p->lambda_; a.u_scale = (T)p->u_scale;
  a.e_scale = (T)(p->noise_rescale == 0.0 ? 1.0 : p->noise_rescale); a.smooth_w = (T)p->smooth_weight;
  a.seed = p->seed; a.call = p->call;
  a.state = (const T*)p->state; a.U = (const T*)p->U; a.u_init = (const T*)p->u_init;
  a.mu = (const T*)p->noise_mu; a.L = (const T*)p->noise_L; a.sinv = (const T*)p->sigma_inv;
  a.umin = (const T*)p->u_min; a.umax = (const T*)p->u_max; a.mp = (const T*)p->model_params;
  a.z = (const T*)p->z; a.sampler = (const T*)p->sampler_actions; a.B = (const T*)p->base_seq;
  a.cost = (T*)p->cost_total; a.omega = (T*)p->omega; a.wnz = (T*)p->cost_total_non_zero;
  a.U_out = (T*)p->U_out; a.action_out = (T*)p->action_out; a.pa = (T*)p->perturbed_action;
  a.noise = (T*)p->noise; a.pert = (T*)p->pert_cost; a.states = (T*)p->states;
  a.record = (T*)p->record;
  T* ws = (T*)p->workspace;
  a.tstamp = nullptr;
  a.M = p->rollout_samples > 1 ? p->rollout_samples : 1;
  a.var_cost = (T)p->rollout_var_cost; a.var_disc = (T)p->rollout_var_discount;
  a.proc_sd = (const T*)p->process_noise_sd;
  a.fuse = -1;
  a.W = a.theta = nullptr; a.S = 0; a.kw = 0; a.kw_jpad = 0;
  // The arrival ticket of the single-launch command sits in the LAST 4 elements of the caller's buffer
  // (workspace + workspace_elems - 4): a place that does not depend on (K, T, nu, num_envs), so that
  // problems of different shapes may share one zero-filled workspace -- at `carve().total - 4` a smaller
  // problem's ticket lay inside a larger problem's scratch, which K1 / K3 overwrite with floats.
  a.ticket = reinterpret_cast<unsigned*>(ws + (p->workspace_elems - 4));
  a.n_env = p->num_envs > 1 ? p->num_envs : 1;
  if (a.n_env > 1 && (p->state_per_sample || p->n_sampler_rows > 0 || p->states != nullptr || p->base_seq != nullptr ||
                      p->S > 0 || p->noise_src == MPPI_NOISE_ACTIONS))
    return fail(MPPI_E_UNSUPPORTED, "num_envs > 1 supports the plain MPPI path only");
  if (a.n_env > 65535) return fail(MPPI_E_BADARG, "num_envs > 65535");
  // per-environment workspace blocks: [n_env][nb1] | [n_env][nkc] | [n_env][nkc][Jpad]
  a.block_min = ws; a.eta_part = ws + (int64_t)a.n_env * c.nb1;
  a.P_part = a.eta_part + (int64_t)a.n_env * c.nkc;
  a.nb1 = c.nb1; a.nkc = c.nkc; a.Jpad = c.Jpad; a.R = c.R;
  return 0;
}

template <typename T>
int need_noise(const KArgs<T>& a) {
  if (a.coloured && a.noise_src != MPPI_NOISE_TNK4) return fail(MPPI_E_BADARG, "noise_coloured needs an external row stream (MPPI_NOISE_TNK4)");
  if (a.noise_src != MPPI_NOISE_PHILOX && a.z == nullptr) return fail(MPPI_E_BADARG, "noise_src needs p->z");
  if (a.noise_src < 0 || a.noise_src > MPPI_NOISE_KTN) return fail(MPPI_E_BADARG, "bad noise_src");
  if (a.noise_src == MPPI_NOISE_KTN &&
      (sizeof(T) != 4 || !a.diag || a.J % 4 != 0 || (reinterpret_cast<uintptr_t>(a.z) & 15) != 0 || a.n_env > 1))
    return fail(MPPI_E_UNSUPPORTED, "MPPI_NOISE_KTN: fp32, diagonal Sigma, T*nu % 4 == 0, 16-byte aligned, single env");
  return 0;
}

template <typename T>
int do_rollout(const MppiProblem* p, hipStream_t st, int fuse = -1, bool kmppi = false, const MppiProblem* theta_problem = nullptr) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (int e = need_noise(a)) return e;
  if (!a.state || !a.cost) return fail(MPPI_E_BADARG, "rollout needs state and cost_total");
  a.fuse = fuse;
  if (kmppi) {
    // interpolation inside K1: z / seed / call describe the support-point stream (S rows of nu)
    if (p->S <= 0 || !p->theta || !p->W) return fail(MPPI_E_BADARG, "mppi_rollout_cost_kmppi needs S, theta, W");
    if (a.noise_src != MPPI_NOISE_TNK4 && a.noise_src != MPPI_NOISE_PHILOX)
      return fail(MPPI_E_UNSUPPORTED, "mppi_rollout_cost_kmppi: support points come from MPPI_NOISE_TNK4 or MPPI_NOISE_PHILOX");
    a.W = (const T*)p->W; a.theta = (const T*)p->theta; a.S = p->S;
    if (theta_problem != nullptr) {
      // mppi_command_kmppi: the kernel may leave the theta update's partial records (one per 256 samples) in the workspace
      // the two problems share, laid out for finalize_blocks on the theta problem
      const MppiProblem* q = theta_problem;
      if (q->K != p->K || q->T != p->S || q->nu != p->nu || q->workspace != p->workspace || q->workspace_elems != p->workspace_elems ||
          q->dtype != p->dtype || q->num_envs > 1)
        return fail(MPPI_E_BADARG, "mppi_command_kmppi: the theta problem must be (K, S, nu) on the trajectory problem's workspace");
      a.kw = 1;
      a.kw_jpad = carve(q).Jpad;
    }
  }
  int r;
  switch (p->model_id) {
    case MPPI_MODEL_PENDULUM: r = rollout_pendulum(a, st); break;
    case MPPI_MODEL_INTEGRATOR: r = rollout_integrator(a, st); break;
    case MPPI_MODEL_LINEAR_GOAL: r = rollout_linear_goal(a, st); break;
    case MPPI_MODEL_MLP: r = rollout_mlp(a, st); break;
    default: {
      const int slot = p->model_id - MPPI_MODEL_CUSTOM_BASE;
      if (slot < 0 || slot >= MAX_CUSTOM) return fail(MPPI_E_UNSUPPORTED, "mppi_rollout_cost: model has no fused kernel");
      const CustomModel& cm = g_custom[slot];
      custom_rollout_fn fn = sizeof(T) == 4 ? cm.f32 : cm.f64;
      if (fn == nullptr || cm.nx != p->nx || cm.nu != p->nu)
        return fail(MPPI_E_UNSUPPORTED, "mppi_rollout_cost: custom model not registered for this (nx, nu, dtype)");
      r = fn((const void*)&a, (void*)st);
    }
  }
  if (r == MPPI_OK_FUSED || r == MPPI_OK_ONCHIP || r == MPPI_OK_KMPPI_W) return r;
  return hipfail(r, "mppi_rollout_cost");
}
}  // namespace

#define BY_DTYPE(p, expr_f32, expr_f64)                                 \
  ((p) == nullptr ? fail(MPPI_E_BADARG, "null problem")                 \
   : (p)->dtype == MPPI_F32 ? (expr_f32)                                \
   : (p)->dtype == MPPI_F64 ? (expr_f64)                                \
                            : fail(MPPI_E_BADARG, "bad dtype"))

extern "C" int mppi_abi_version(void) { return MPPI_ABI_VERSION; }
extern "C" int64_t mppi_problem_size(void) { return (int64_t)sizeof(MppiProblem); }
extern "C" const char* mppi_last_error(void) { return g_err; }

extern "C" int64_t mppi_workspace_elems(const MppiProblem* p) {
  if (p == nullptr || p->K <= 0 || p->T <= 0 || p->nu <= 0) return 0;
  return carve(p).total;
}

extern "C" int mppi_model_supported(int32_t model_id, int32_t nx, int32_t nu, int32_t dtype, int32_t hidden) {
  if (dtype != MPPI_F32 && dtype != MPPI_F64) return 0;
  switch (model_id) {
    case MPPI_MODEL_PENDULUM: return supported_pendulum(nx, nu, hidden);
    case MPPI_MODEL_INTEGRATOR: return supported_integrator(nx, nu, hidden);
    case MPPI_MODEL_LINEAR_GOAL: return supported_linear_goal(nx, nu, hidden);
    case MPPI_MODEL_MLP: return supported_mlp(nx, nu, hidden);
    default: {
      const int slot = model_id - MPPI_MODEL_CUSTOM_BASE;
      if (slot < 0 || slot >= MAX_CUSTOM) return 0;
      const CustomModel& cm = g_custom[slot];
      return cm.nx == nx && cm.nu == nu && (dtype == MPPI_F32 ? cm.f32 != nullptr : cm.f64 != nullptr);
    }
  }
}

extern "C" int mppi_register_model(int32_t model_id, int32_t nx, int32_t nu, void* f32, void* f64) {
  const int slot = model_id - MPPI_MODEL_CUSTOM_BASE;
  if (slot < 0 || slot >= MAX_CUSTOM) return fail(MPPI_E_BADARG, "custom model id out of range");
  if (nx <= 0 || nu <= 0 || (f32 == nullptr && f64 == nullptr)) return fail(MPPI_E_BADARG, "bad custom model");
  g_custom[slot] = CustomModel{nx, nu, (custom_rollout_fn)f32, (custom_rollout_fn)f64};
  return 0;
}

template <typename T>
static int do_fill(const MppiProblem* p, void* z, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!z) return fail(MPPI_E_BADARG, "null z");
  return hipfail(launch_noise_fill_philox<T>(a, (T*)z, st), "mppi_noise_fill_philox");
}
extern "C" int mppi_noise_fill_philox(const MppiProblem* p, void* z, void* stream) {
  return BY_DTYPE(p, do_fill<float>(p, z, (hipStream_t)stream), do_fill<double>(p, z, (hipStream_t)stream));
}

template <typename T>
static int do_fill_coloured(const MppiProblem* p, void* z, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!z) return fail(MPPI_E_BADARG, "null eps");
  return hipfail(launch_noise_fill_philox_coloured<T>(a, (T*)z, st), "mppi_noise_fill_philox_coloured");
}
extern "C" int mppi_noise_fill_philox_coloured(const MppiProblem* p, void* z, void* stream) {
  return BY_DTYPE(p, do_fill_coloured<float>(p, z, (hipStream_t)stream),
                  do_fill_coloured<double>(p, z, (hipStream_t)stream));
}

template <typename T>
static int do_from_ktn(const MppiProblem* p, const void* in, void* out, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!in || !out) return fail(MPPI_E_BADARG, "null z");
  return hipfail(launch_noise_from_ktn<T>(a, (const T*)in, (T*)out, st), "mppi_noise_from_ktn");
}
extern "C" int mppi_noise_from_ktn(const MppiProblem* p, const void* in, void* out, void* stream) {
  return BY_DTYPE(p, do_from_ktn<float>(p, in, out, (hipStream_t)stream),
                  do_from_ktn<double>(p, in, out, (hipStream_t)stream));
}

template <typename T>
static int do_interp(const MppiProblem* p, void* out, hipStream_t st) {
  // p describes the trajectory problem (T, U); theta/W/S describe the support points
  if (p == nullptr || p->S <= 0 || !p->theta || !p->W || !out) return fail(MPPI_E_BADARG, "kmppi_interp needs S, theta, W");
  MppiProblem q = *p;
  q.T = p->S; q.U = p->theta; q.shift = 0; q.sample_null_action = 0; q.n_sampler_rows = 0;
  q.base_seq = nullptr; q.noise_rescale = 1.0; q.smooth_weight = 0.0;
  KArgs<T> a;
  if (int e = make_args<T>(&q, a)) return e;
  if (int e = need_noise(a)) return e;
  if (a.noise_src == MPPI_NOISE_ACTIONS) return fail(MPPI_E_BADARG, "kmppi_interp needs a noise stream");
  const int J4out = (int)mppi_noise_rows4(p->T, p->nu);
  return hipfail(launch_kmppi_interp<T>(a, (const T*)p->W, p->T, J4out, (T*)out, st), "mppi_kmppi_interp");
}
extern "C" int mppi_kmppi_interp(const MppiProblem* p, void* out, void* stream) {
  return BY_DTYPE(p, do_interp<float>(p, out, (hipStream_t)stream), do_interp<double>(p, out, (hipStream_t)stream));
}

extern "C" int mppi_kmppi_shift(int32_t dtype, int32_t T, int32_t S, int32_t nu, const void* U, const void* u_init,
                                const void* theta, const void* W_shift, void* U_out, void* theta_out, void* stream) {
  if (T <= 0 || S <= 0 || nu <= 0 || !U || !u_init || !theta || !W_shift || !U_out || !theta_out)
    return fail(MPPI_E_BADARG, "mppi_kmppi_shift: bad argument");
  if (dtype == MPPI_F32)
    return hipfail(launch_kmppi_sequences<float>(S, S, nu, (const float*)W_shift, (const float*)theta, (float*)theta_out, T,
                                                 (const float*)U, (const float*)u_init, (float*)U_out, (hipStream_t)stream), "mppi_kmppi_shift");
  if (dtype == MPPI_F64)
    return hipfail(launch_kmppi_sequences<double>(S, S, nu, (const double*)W_shift, (const double*)theta, (double*)theta_out, T,
                                                  (const double*)U, (const double*)u_init, (double*)U_out, (hipStream_t)stream), "mppi_kmppi_shift");
  return fail(MPPI_E_BADARG, "bad dtype");
}
// ---- small host -> device upload through the kernel arguments -----------------------------------------
// The state of a closed loop arrives from the host every step (nx values).  A pageable hipMemcpy costs the host
// ~20 us, pinned staging needs an event per slot; here the bytes travel INSIDE the launch packet (kernarg) of a
// one-wave kernel that writes them to `dst`: the host buffer is consumed before the call returns, nothing to
// keep alive, nothing to wait for.
namespace {
template <int WORDS> struct UploadBlob { unsigned w[WORDS]; };
template <int WORDS>
__global__ void __launch_bounds__(64) upload_small_kernel(const UploadBlob<WORDS> b, unsigned* __restrict__ dst, int nwords) {
  for (int i = threadIdx.x; i < nwords; i += 64) dst[i] = b.w[i];
}
template <int WORDS>
int upload_small(const void* src, int64_t nbytes, void* dst, hipStream_t st) {
  UploadBlob<WORDS> b;
  memcpy(b.w, src, (size_t)nbytes);
  hipLaunchKernelGGL(upload_small_kernel<WORDS>, dim3(1), dim3(64), 0, st, b, (unsigned*)dst, (int)(nbytes / 4));
  return (int)hipGetLastError();
}
}  // namespace
extern "C" int mppi_upload_small(const void* src_host, int64_t nbytes, void* dst_device, void* stream) {
  if (src_host == nullptr || dst_device == nullptr || nbytes <= 0 || nbytes % 4 != 0 || nbytes > 2048)
    return fail(MPPI_E_BADARG, "mppi_upload_small: 4..2048 bytes, a multiple of 4");
  if (nbytes <= 128) return hipfail(upload_small<32>(src_host, nbytes, dst_device, (hipStream_t)stream), "mppi_upload_small");
  return hipfail(upload_small<512>(src_host, nbytes, dst_device, (hipStream_t)stream), "mppi_upload_small");
}

// ---- test / debug seam: the process-noise normals of the fused multi-rollout K1 ---------------------------
namespace {
template <typename T>
__global__ void __launch_bounds__(BLOCK) process_noise_export_kernel(unsigned long long seed, unsigned long long call, long long k_offset,
                                                                     int K, int Tn, int M, int nx, int seven, T* __restrict__ out) {
  const int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= K) return;
  const int nxb = (nx + 3) / 4;
  for (int t = blockIdx.y; t < Tn; t += gridDim.y)
    for (int m = 0; m < M; ++m)
      for (int q = 0; q < nxb; ++q) {
        T w[4];
        philox_normal4<T>(seed ^ PROCESS_NOISE_KEY_TAG, call, k_offset + k, ((long long)t * PROCESS_NOISE_MM + m) * nxb + q, w, seven != 0);
        for (int i = 0; i < 4; ++i)
          if (4 * q + i < nx) out[(((long long)m * K + k) * Tn + t) * nx + 4 * q + i] = w[i];
      }
}
}  // namespace
extern "C" int mppi_process_noise_export(const MppiProblem* p, void* out, void* stream) {
  if (p == nullptr || out == nullptr || p->K <= 0 || p->T <= 0 || p->nx <= 0) return fail(MPPI_E_BADARG, "mppi_process_noise_export: bad argument");
  const int M = p->rollout_samples > 1 ? p->rollout_samples : 1;
  if (M > PROCESS_NOISE_MM) return fail(MPPI_E_UNSUPPORTED, "mppi_process_noise_export: the fused stream is defined for M <= 4");
  const dim3 grid((p->K + BLOCK - 1) / BLOCK, p->T < 64 ? p->T : 64);
  if (p->dtype == MPPI_F32)
    hipLaunchKernelGGL(process_noise_export_kernel<float>, grid, dim3(BLOCK), 0, (hipStream_t)stream, p->seed, p->call, p->k_offset,
                       p->K, p->T, M, p->nx, p->philox_rounds == 7 ? 1 : 0, (float*)out);
  else if (p->dtype == MPPI_F64)
    hipLaunchKernelGGL(process_noise_export_kernel<double>, grid, dim3(BLOCK), 0, (hipStream_t)stream, p->seed, p->call, p->k_offset,
                       p->K, p->T, M, p->nx, p->philox_rounds == 7 ? 1 : 0, (double*)out);
  else
    return fail(MPPI_E_BADARG, "bad dtype");
  return hipfail((int)hipGetLastError(), "mppi_process_noise_export");
}

extern "C" int mppi_smppi_shift(int32_t dtype, int32_t T, int32_t nu, const void* U, const void* u_init, const void* A, double dt,
                                void* U_out, void* A_out, void* B_out, void* stream) {
  if (T <= 0 || nu <= 0 || !U || !u_init || !A || !U_out || !A_out || !B_out) return fail(MPPI_E_BADARG, "mppi_smppi_shift: bad argument");
  if (dtype == MPPI_F32)
    return hipfail(launch_smppi_shift<float>(T, nu, (const float*)U, (const float*)u_init, (const float*)A, (float)dt, (float*)U_out,
                                             (float*)A_out, (float*)B_out, (hipStream_t)stream), "mppi_smppi_shift");
  if (dtype == MPPI_F64)
    return hipfail(launch_smppi_shift<double>(T, nu, (const double*)U, (const double*)u_init, (const double*)A, dt, (double*)U_out,
                                              (double*)A_out, (double*)B_out, (hipStream_t)stream), "mppi_smppi_shift");
  return fail(MPPI_E_BADARG, "bad dtype");
}
extern "C" int mppi_kmppi_trajectory(int32_t dtype, int32_t T, int32_t S, int32_t nu, const void* W, const void* theta,
                                     void* U_out, void* stream) {
  if (T <= 0 || S <= 0 || nu <= 0 || !W || !theta || !U_out) return fail(MPPI_E_BADARG, "mppi_kmppi_trajectory: bad argument");
  if (dtype == MPPI_F32)
    return hipfail(launch_kmppi_sequences<float>(T, S, nu, (const float*)W, (const float*)theta, (float*)U_out, 0, nullptr, nullptr,
                                                 nullptr, (hipStream_t)stream), "mppi_kmppi_trajectory");
  if (dtype == MPPI_F64)
    return hipfail(launch_kmppi_sequences<double>(T, S, nu, (const double*)W, (const double*)theta, (double*)U_out, 0, nullptr,
                                                  nullptr, nullptr, (hipStream_t)stream), "mppi_kmppi_trajectory");
  return fail(MPPI_E_BADARG, "bad dtype");
}

extern "C" int mppi_kmppi_after_update(int32_t dtype, int32_t T, int32_t S, int32_t nu, const void* W, const void* W_shift, const void* theta,
                                       const void* u_init, void* U_out, void* theta_shift_out, void* U_shift_out, void* stream) {
  if (T <= 0 || S <= 0 || nu <= 0 || !W || !W_shift || !theta || !u_init || !U_out || !theta_shift_out || !U_shift_out)
    return fail(MPPI_E_BADARG, "mppi_kmppi_after_update: bad argument");
  if (dtype == MPPI_F32)
    return hipfail(launch_kmppi_after_update<float>(T, S, nu, (const float*)W, (const float*)W_shift, (const float*)theta, (const float*)u_init,
                                                    (float*)U_out, (float*)theta_shift_out, (float*)U_shift_out, (hipStream_t)stream), "mppi_kmppi_after_update");
  if (dtype == MPPI_F64)
    return hipfail(launch_kmppi_after_update<double>(T, S, nu, (const double*)W, (const double*)W_shift, (const double*)theta, (const double*)u_init,
                                                     (double*)U_out, (double*)theta_shift_out, (double*)U_shift_out, (hipStream_t)stream), "mppi_kmppi_after_update");
  return fail(MPPI_E_BADARG, "bad dtype");
}

extern "C" int mppi_rollout_cost(const MppiProblem* p, void* stream) {
  return BY_DTYPE(p, do_rollout<float>(p, (hipStream_t)stream), do_rollout<double>(p, (hipStream_t)stream));
}
static std::atomic<long long> g_kmppi_fused_rollouts{0};   // controllers may be driven from several host threads
extern "C" int64_t mppi_stat_kmppi_fused_rollouts(void) { return g_kmppi_fused_rollouts.load(); }
extern "C" int mppi_rollout_cost_kmppi(const MppiProblem* p, void* stream) {
  const int r = BY_DTYPE(p, do_rollout<float>(p, (hipStream_t)stream, -1, true), do_rollout<double>(p, (hipStream_t)stream, -1, true));
  if (r == 0) ++g_kmppi_fused_rollouts;
  return r;
}

template <typename T>
static int do_prepare(const MppiProblem* p, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (int e = need_noise(a)) return e;
  return hipfail(launch_prepare<T>(a, st), "mppi_prepare");
}
extern "C" int mppi_prepare(const MppiProblem* p, void* stream) {
  return BY_DTYPE(p, do_prepare<float>(p, (hipStream_t)stream), do_prepare<double>(p, (hipStream_t)stream));
}

template <typename T>
static int do_bmin(const MppiProblem* p, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!a.cost) return fail(MPPI_E_BADARG, "null cost_total");
  return hipfail(launch_cost_block_min<T>(a, st), "mppi_cost_block_min");
}
extern "C" int mppi_cost_block_min(const MppiProblem* p, void* stream) {
  return BY_DTYPE(p, do_bmin<float>(p, (hipStream_t)stream), do_bmin<double>(p, (hipStream_t)stream));
}

static thread_local int t_next_draw = 0;
extern "C" int mppi_last_next_draw(void) { return t_next_draw; }
// fp32 rows in the engine's layout, diagonal Sigma: ONE kernel with or without the next draw (noise_torch.hip), so that a command's
// bits do not depend on whether its K3 launch also generated
template <typename T>
static int weights_rows(const KArgs<T>&, void*, const MppiProblem*, hipStream_t) { return MPPI_E_UNSUPPORTED; }
template <>
int weights_rows<float>(const KArgs<float>& a, void* next_z, const MppiProblem* p, hipStream_t st) {
  return launch_weights_partial_rows_f32(a, next_z, p->next_kind, p->next_seed, p->next_philox_offset, p->next_grid_blocks, st);
}
template <typename T>
static int do_weights(const MppiProblem* p, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (int e = need_noise(a)) return e;
  if (!a.cost) return fail(MPPI_E_BADARG, "null cost_total");
  t_next_draw = 0;
  static const bool off = getenv("MPPI_NO_NEXT_DRAW") != nullptr;       // A/B knob for tools/, read once
  if (p->next_z != nullptr && !off) {
    // ABI 21: the next command's draw beside this K3 where that launch exists
    const int e = weights_rows<T>(a, p->next_z, p, st);
    if (e == 0) { t_next_draw = 1; return 0; }
    if (e != MPPI_E_UNSUPPORTED) return hipfail(e, "mppi_weights_partial (with the next draw)");
  }
  const int e = weights_rows<T>(a, nullptr, p, st);
  if (e != MPPI_E_UNSUPPORTED) return hipfail(e, "mppi_weights_partial");
  return hipfail(launch_weights_partial<T>(a, st), "mppi_weights_partial");
}
extern "C" int mppi_weights_partial(const MppiProblem* p, void* stream) {
  return BY_DTYPE(p, do_weights<float>(p, (hipStream_t)stream), do_weights<double>(p, (hipStream_t)stream));
}

template <typename T>
static int do_finalize(const MppiProblem* p, int apply, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!a.record || !a.cost) return fail(MPPI_E_BADARG, "finalize needs record and cost_total");
  if (apply && !a.U_out) return fail(MPPI_E_BADARG, "finalize(apply) needs U_out");
  return hipfail(launch_finalize<T>(a, apply, st), "mppi_finalize");
}
extern "C" int mppi_finalize(const MppiProblem* p, int apply, void* stream) {
  return BY_DTYPE(p, do_finalize<float>(p, apply, (hipStream_t)stream),
                  do_finalize<double>(p, apply, (hipStream_t)stream));
}

static std::atomic<long long> g_single_launch_commands{0};
extern "C" int64_t mppi_stat_single_launch_commands(void) { return g_single_launch_commands.load(); }
static std::atomic<long long> g_onchip_commands{0};
static std::atomic<long long> g_kmppi_onchip_updates{0};
extern "C" int64_t mppi_stat_onchip_commands(void) { return g_onchip_commands.load(); }
extern "C" int64_t mppi_stat_mlp_split_launches(void) { return (int64_t)mlp_split_launches(); }
static std::atomic<long long> g_onchip_pair_launches{0};
namespace mppi { void onchip_pair_launched() { ++g_onchip_pair_launches; } }
extern "C" int64_t mppi_stat_onchip_pair_launches(void) { return g_onchip_pair_launches.load(); }
// which form the calling THREAD's last mppi_command took (the counters above are process-wide: with two controllers
// commanding from two threads, "did MY command run on chip" cannot be read off a shared count -- ADVICE r03)
static thread_local int t_last_form = MPPI_FORM_NONE;
extern "C" int mppi_last_command_form(void) { return t_last_form; }

template <typename T>
static int do_finalize_blocks(const MppiProblem* p, int apply, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!a.record || !a.cost) return fail(MPPI_E_BADARG, "finalize needs record and cost_total");
  if (apply && !a.U_out) return fail(MPPI_E_BADARG, "finalize(apply) needs U_out");
  onchip_carve(a);                       // one partial record per 256-sample workgroup of the on-chip K1
  return hipfail(launch_finalize_blocks<T>(a, apply, st), "mppi_command (on-chip finalize)");
}

extern "C" int mppi_command_kmppi(const MppiProblem* p, const MppiProblem* theta_problem, int apply, void* stream) {
  if (theta_problem == nullptr) return fail(MPPI_E_BADARG, "mppi_command_kmppi: null theta problem");
  t_next_draw = 0;
  const int r = BY_DTYPE(p, do_rollout<float>(p, (hipStream_t)stream, -1, true, theta_problem),
                         do_rollout<double>(p, (hipStream_t)stream, -1, true, theta_problem));
  if (r == MPPI_OK_KMPPI_W) {
    // K1 interpolated, rolled out and reduced its workgroups' part of the theta update from the control points it
    // still held: this launch combines the K/256 partial records in workgroup order and applies the update
    ++g_kmppi_fused_rollouts;
    ++g_kmppi_onchip_updates;
    return BY_DTYPE(theta_problem, do_finalize_blocks<float>(theta_problem, apply, (hipStream_t)stream),
                    do_finalize_blocks<double>(theta_problem, apply, (hipStream_t)stream));
  }
  if (r != 0) return r;               // MPPI_E_UNSUPPORTED: no fused-interpolation kernel here -- the caller's two-launch form
  ++g_kmppi_fused_rollouts;
  if (int e = mppi_weights_partial(theta_problem, stream)) return e;
  return mppi_finalize(theta_problem, apply, stream);
}
extern "C" int64_t mppi_stat_kmppi_onchip_updates(void) { return g_kmppi_onchip_updates.load(); }

extern "C" int64_t mppi_onchip_spill_elems(const MppiProblem* p) {
  if (p == nullptr || p->dtype != MPPI_F32 || p->K <= 0 || p->T <= 0 || p->nu <= 0) return 0;
  const mppi::OnChipGeometry g = mppi::onchip_geometry(p->nu, p->T, p->sigma_diagonal != 0);
  if (!g.ok || g.nsm <= 0) return 0;
  const int64_t nkc = (p->K + mppi::BLOCK - 1) / mppi::BLOCK;
  int64_t n = (int64_t)g.nsm * g.P4 * nkc * mppi::BLOCK * 4;
  // the two-waves-per-sample form (rollout_onchip_pair.hpp) keeps fewer rows in registers: its array is the larger one
  // (the models it is instantiated for: rollout_onchip_pair.hpp onchip_pair_model_ok)
  const bool pair_model = p->model_id == MPPI_MODEL_INTEGRATOR && p->nx == 16 && p->nu == 12;
  for (int plain = 0; plain < 2; ++plain) {     // (the SMPPI terms' hand-over is larger: LDS may hold a tile fewer)
    const mppi::OnChipPairGeometry gp = mppi::onchip_pair_geometry(p->nu, p->nx, p->T, plain != 0);
    if (pair_model && gp.ok && gp.nch >= 4 && p->sigma_diagonal != 0) {
      const int64_t np = (int64_t)gp.nsm * gp.P4 * nkc * 2 * mppi::BLOCK * 4;
      if (np > n) n = np;
    }
  }
  return n;
}

extern "C" int mppi_command(const MppiProblem* p, int apply, void* stream) {
  t_next_draw = 0;
  // small problems: K1's launch carries K3 and K4 as well when the caller left omega and
  // cost_total_non_zero NULL (they are functions of cost_total and the record: see the header)
  const int e1 = BY_DTYPE(p, do_rollout<float>(p, (hipStream_t)stream, apply ? 1 : 0),
                          do_rollout<double>(p, (hipStream_t)stream, apply ? 1 : 0));
  t_last_form = MPPI_FORM_NONE;
  if (e1 == MPPI_OK_FUSED) { ++g_single_launch_commands; t_last_form = MPPI_FORM_SINGLE_LAUNCH; return 0; }
  if (e1 == MPPI_OK_ONCHIP) {
    t_last_form = MPPI_FORM_ONCHIP;
    // rng = engine generator, no row array (p->z == NULL): K1 generated, rolled out, kept the bounded noise on
    // chip and left one partial record per workgroup (csrc/rollout_onchip.hpp); this launch combines them
    ++g_onchip_commands;
    return BY_DTYPE(p, do_finalize_blocks<float>(p, apply, (hipStream_t)stream), do_finalize_blocks<double>(p, apply, (hipStream_t)stream));
  }
  if (e1) return e1;
  MppiProblem q = *p;
  // "generate once": K1 stored the Philox rows it generated, K3 re-reads them
  if (q.noise_src == MPPI_NOISE_PHILOX && q.z != nullptr) q.noise_src = MPPI_NOISE_TNK4;
  t_last_form = MPPI_FORM_STREAMING;
  if (int e = mppi_weights_partial(&q, stream)) return e;
  return mppi_finalize(&q, apply, stream);
}

template <typename T>
static int do_combine(const MppiProblem* p, const void* rec, int G, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!rec || G <= 0 || !a.U_out || !a.cost) return fail(MPPI_E_BADARG, "combine needs records, U_out, cost_total");
  return hipfail(launch_combine<T>(a, (const T*)rec, G, st), "mppi_combine");
}
extern "C" int mppi_combine(const MppiProblem* p, const void* rec, int32_t G, void* stream) {
  return BY_DTYPE(p, do_combine<float>(p, rec, G, (hipStream_t)stream),
                  do_combine<double>(p, rec, G, (hipStream_t)stream));
}

template <typename T>
static int do_combine_ptrs(const MppiProblem* p, const void* const* ptrs, int G, hipStream_t st) {
  KArgs<T> a;
  if (int e = make_args<T>(p, a)) return e;
  if (!ptrs || G <= 0 || G > MPPI_MAX_GROUP || !a.U_out || !a.cost) return fail(MPPI_E_BADARG, "combine needs 1..MPPI_MAX_GROUP record pointers, U_out, cost_total");
  for (int g = 0; g < G; ++g)
    if (ptrs[g] == nullptr) return fail(MPPI_E_BADARG, "mppi_combine_ptrs: null record pointer");
  return hipfail(launch_combine<T>(a, (const T*)nullptr, G, st, (const T* const*)ptrs), "mppi_combine_ptrs");
}
extern "C" int mppi_combine_ptrs(const MppiProblem* p, const void* const* record_ptrs, int32_t G, void* stream) {
  return BY_DTYPE(p, do_combine_ptrs<float>(p, record_ptrs, G, (hipStream_t)stream),
                  do_combine_ptrs<double>(p, record_ptrs, G, (hipStream_t)stream));
}
