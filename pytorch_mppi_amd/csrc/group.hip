// group.hip -- one process, N devices: a whole sharded command issued from ONE place (ABI 22).
//
// The reference's caller is one Python process stepping one environment (mppi.py:876-898).  `MPPI(..., devices=[...])` keeps
// that shape, but until ABI 21 the shards' launches were issued one after the other by the one Python thread (~40 us of host
// time per shard): commands shorter than N x 40 us were bound by the host, not by the GPUs.  Here every device of the group
// has a worker thread inside the library -- its device current once, for good -- and a command is
//
//     caller:   mppi_group_broadcast (optional)   state on devs[0]  -> every other device, in front of its K1
//               mppi_group_submit x N             the problem block of device g, handed over as soon as it is filled
//               mppi_group_wait                   until every worker has ISSUED its part (no device synchronisation)
//     worker g: K1, K3, K4 (record only)          mppi_command(apply = 0) / mppi_command_kmppi
//               exchange                          RCCL: ncclAllGather on its own communicator (one thread per device: no group
//                                                 call needed); staged (a device listed twice, no RCCL): an event behind K4,
//                                                 a host barrier, this stream waits for the other shards' events
//               K5                                the rank-order combine -> bit-identical U on every device (staged: reading
//                                                 the records where they are, mppi_combine_ptrs)
//
// so the host's share of a command is the caller's N hand-overs (a copy of the block each) plus ONE device's launches.
// Nothing here allocates device memory or synchronises a device; events are created with the workers.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "common.hpp"

int mppi_fail_message(int code, const char* msg);      // capi.hip
namespace mppi_dist {                                   // dist.hip
int all_gather_record(const MppiProblem* q, void* comm, void* records, void* stream);
}

namespace {
constexpr int SPIN_BEFORE_SLEEP = 1 << 15;              // ~100-200 us of polling behind a command, then the worker sleeps

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

struct Group;
struct Slot {
  Group* group = nullptr;
  int index = 0, dev = 0;
  void* comm = nullptr;
  std::thread th;
  // mailbox: the caller bumps `submitted`, the worker answers with `done`
  std::atomic<uint32_t> submitted{0}, done{0};
  std::atomic<int> sleeping{0};
  uint64_t cmd = 0;                             // the group command this hand-over belongs to
  MppiProblem p, pt;
  bool has_pt = false;
  void* records = nullptr;
  hipStream_t stream = nullptr;
  // broadcast of the state in front of this command's K1 (device 0 -> this device)
  const void* bc_src = nullptr;
  void* bc_dst = nullptr;
  int64_t bc_bytes = 0;
  // results of the last command
  int rc = 0, form = 0, next_draw = 0;
  char err[256] = "";
  hipEvent_t ev[2] = {nullptr, nullptr};      // "my record is written" -- two, by command parity (see exchange)
  int init_rc = 0;
  bool peers_ok = true;                         // this device can read every other device's memory (peer access enabled)
};

struct Group {
  int n = 0;
  bool staged = true;
  std::vector<Slot*> slots;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> stop{false};
  uint64_t cmd = 1;                             // the command being assembled (caller's side)
  int pending = 0;                              // hand-overs of it so far
  // (cmd << 1) | aborted of the last command the caller has DECIDED: its workers go on to the exchange (all N parts are in) or
  // drop it (mppi_group_abort: a part is missing -- an all-gather without it would leave the other devices waiting on the GPU)
  std::atomic<uint64_t> decided{0};
  // host barrier of the workers inside a committed command
  std::atomic<int> arrived{0};
  std::atomic<uint32_t> generation{0};
  std::atomic<int> failed{0};                   // a worker's K1 / K3 / K4 failed: nobody exchanges
  hipEvent_t bc_ev = nullptr;                   // "the state on devs[0] is ready" (recorded by the caller on devs[0]'s stream)
  bool bc_armed = false;
  bool in_place = false;                        // staged exchange: K5 reads the records where they are (mppi_combine_ptrs)
};

void barrier(Group* g) {
  const uint32_t gen = g->generation.load(std::memory_order_acquire);
  if (g->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == g->n) {
    g->arrived.store(0, std::memory_order_relaxed);
    g->generation.store(gen + 1, std::memory_order_release);
    return;
  }
  while (g->generation.load(std::memory_order_acquire) == gen) cpu_relax();
}

int keep_error(Slot* s, int rc, const char* msg = nullptr) {
  s->rc = rc;
  snprintf(s->err, sizeof(s->err), "device %d: %s", s->dev, msg ? msg : mppi_last_error());
  return rc;
}
int keep_hip_error(Slot* s, hipError_t e) { return keep_error(s, (int)e, hipGetErrorString(e)); }

// The exchange of a committed command + K5.  RCCL: ncclAllGather on this device's communicator (one thread per device: every
// worker calls the collective itself, no group call), then mppi_combine on the gathered records.  Staged: an event behind this
// device's K4; this device's stream waits for the other devices' events and K5 reads their records where they are
// (mppi_combine_ptrs: same device, or peers with access enabled) -- or, without peer access, after copies into `records`.
// Two events per device, used by command parity: command n + 2 re-records the event of command n, and by then every other
// stream has passed its wait on it (their command n + 1 waited for this device's command-(n + 1) event, recorded behind this
// device's command-n reads) -- the same chain makes it safe for the CALLER to recycle a record buffer two commands later, which
// is what a caching allocator does with it.
// Either way nobody starts before EVERY device has issued its K1 / K3 / K4 without error (host barrier): a collective that one
// rank never joins leaves the others spinning on the GPU.
int exchange_combine(Slot* s, const MppiProblem* q, int rc_so_far) {
  Group* g = s->group;
  const int par = (int)(s->cmd & 1u);
  hipError_t e = hipSuccess;
  if (rc_so_far == 0 && g->staged) e = hipEventRecord(s->ev[par], s->stream);
  if (rc_so_far != 0 || e != hipSuccess) g->failed.store(1, std::memory_order_release);
  barrier(g);
  if (e != hipSuccess) return keep_hip_error(s, e);
  if (rc_so_far != 0) return rc_so_far;
  if (g->failed.load(std::memory_order_acquire)) return keep_error(s, MPPI_E_GROUP_PEER, "another device of the group failed to issue its part of the command");
  if (!g->staged) {
    int r = mppi_dist::all_gather_record(q, s->comm, s->records, s->stream);
    if (r == 0) r = mppi_combine(q, s->records, g->n, s->stream);
    return r != 0 ? keep_error(s, r) : 0;
  }
  const size_t n = 2 + (size_t)q->T * q->nu;
  const size_t bytes = n * (q->dtype == MPPI_F64 ? 8 : 4);
  const void* ptrs[MPPI_MAX_GROUP];
  for (int j = 0; j < g->n; ++j) {
    Slot* o = g->slots[j];
    const MppiProblem* oq = o->has_pt ? &o->pt : &o->p;
    if (j != s->index && (o->stream != s->stream || o->dev != s->dev)) {
      e = hipStreamWaitEvent(s->stream, o->ev[par], 0);
      if (e != hipSuccess) return keep_hip_error(s, e);
    }
    ptrs[j] = oq->record;
    if (g->in_place) continue;
    char* dst = (char*)s->records + (size_t)j * bytes;
    if (o->dev == s->dev) e = hipMemcpyAsync(dst, oq->record, bytes, hipMemcpyDeviceToDevice, s->stream);
    else e = hipMemcpyPeerAsync(dst, s->dev, oq->record, o->dev, bytes, s->stream);
    if (e != hipSuccess) return keep_hip_error(s, e);
  }
  const int r = g->in_place ? mppi_combine_ptrs(q, ptrs, g->n, s->stream) : mppi_combine(q, s->records, g->n, s->stream);
  return r != 0 ? keep_error(s, r) : 0;
}

void run_command(Slot* s) {
  Group* g = s->group;
  s->rc = 0; s->form = 0; s->next_draw = 0; s->err[0] = 0;
  int rc = 0;
  if (s->bc_bytes > 0 && s->bc_dst != nullptr && s->bc_dst != s->bc_src) {
    hipError_t e = hipStreamWaitEvent(s->stream, g->bc_ev, 0);
    if (e == hipSuccess) {
      const int src_dev = g->slots[0]->dev;
      e = src_dev == s->dev ? hipMemcpyAsync(s->bc_dst, s->bc_src, (size_t)s->bc_bytes, hipMemcpyDeviceToDevice, s->stream)
                            : hipMemcpyPeerAsync(s->bc_dst, s->dev, s->bc_src, src_dev, (size_t)s->bc_bytes, s->stream);
    }
    if (e != hipSuccess) rc = keep_hip_error(s, e);
  }
  if (rc == 0) {
    rc = s->has_pt ? mppi_command_kmppi(&s->p, &s->pt, /*apply=*/0, s->stream) : mppi_command(&s->p, /*apply=*/0, s->stream);
    s->form = mppi_last_command_form();
    s->next_draw = mppi_last_next_draw();
    if (rc != 0) keep_error(s, rc);
  }
  // K1 / K3 / K4 are on their way while the caller still fills the other devices' blocks; the exchange needs its decision
  uint64_t d;
  while (((d = g->decided.load(std::memory_order_acquire)) >> 1) != s->cmd) cpu_relax();
  if (d & 1u) {
    if (rc == 0) keep_error(s, MPPI_E_GROUP_PEER, "the command was abandoned before every device had its part (mppi_group_abort)");
    return;
  }
  // the record of the exchange: the theta problem's for KMPPI; K1 stored the Philox rows it generated (mppi_command_sharded)
  MppiProblem q = s->has_pt ? s->pt : s->p;
  if (q.noise_src == MPPI_NOISE_PHILOX && q.z != nullptr) q.noise_src = MPPI_NOISE_TNK4;
  (void)exchange_combine(s, &q, rc);
}

void worker(Slot* s) {
  Group* g = s->group;
  hipError_t e = hipSetDevice(s->dev);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming);
  if (e == hipSuccess && s->index == 0) e = hipEventCreateWithFlags(&g->bc_ev, hipEventDisableTiming);
  // staged exchange: may this device read the other devices' records in place?
  for (Slot* o : g->slots) {
    if (e != hipSuccess || o->dev == s->dev) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, s->dev, o->dev) != hipSuccess || !can) { s->peers_ok = false; (void)hipGetLastError(); continue; }
    const hipError_t pe = hipDeviceEnablePeerAccess(o->dev, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) s->peers_ok = false;
    (void)hipGetLastError();                                  // ("already enabled" is sticky otherwise)
  }
  s->init_rc = (int)e;
  s->done.store(1, std::memory_order_release);              // "initialised" (submitted starts at 1 too: see mppi_group_create)
  uint32_t seen = 1;
  for (;;) {
    int spins = 0;
    while (s->submitted.load(std::memory_order_acquire) == seen && !g->stop.load(std::memory_order_acquire)) {
      if (++spins < SPIN_BEFORE_SLEEP) { cpu_relax(); continue; }
      std::unique_lock<std::mutex> lk(g->mu);
      s->sleeping.store(1, std::memory_order_seq_cst);
      g->cv.wait(lk, [&] { return s->submitted.load(std::memory_order_acquire) != seen || g->stop.load(std::memory_order_acquire); });
      s->sleeping.store(0, std::memory_order_seq_cst);
    }
    if (g->stop.load(std::memory_order_acquire)) break;
    seen = s->submitted.load(std::memory_order_acquire);
    run_command(s);
    s->done.store(seen, std::memory_order_release);
  }
  for (int i = 0; i < 2; ++i)
    if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
  if (s->index == 0 && g->bc_ev) (void)hipEventDestroy(g->bc_ev);
}

// decide the command being assembled (commit: every part is in; abort: drop it) and wait until its workers have finished issuing
int finish(Group* g, bool abort, int32_t* forms, int32_t* next_draws) {
  g->decided.store((g->cmd << 1) | (abort ? 1u : 0u), std::memory_order_release);
  int rc = 0;
  for (int i = 0; i < g->n; ++i) {
    Slot* s = g->slots[i];
    if (s->cmd != g->cmd) continue;                          // (abort: this device never got its part)
    const uint32_t want = s->submitted.load(std::memory_order_relaxed);
    while (s->done.load(std::memory_order_acquire) != want) cpu_relax();
    if (forms) forms[i] = s->form;
    if (next_draws) next_draws[i] = s->next_draw;
    // the first device's own failure, not the "a peer failed" of the others
    if (s->rc != 0 && (rc == 0 || (rc == MPPI_E_GROUP_PEER && s->rc != MPPI_E_GROUP_PEER))) { rc = s->rc; mppi_fail_message(rc, s->err); }
  }
  ++g->cmd;
  g->pending = 0;
  return rc;
}
}  // namespace

extern "C" int mppi_group_create(int32_t ndev, const int32_t* devs, void* const* comms, void** group_out) {
  if (ndev <= 0 || ndev > MPPI_MAX_GROUP || devs == nullptr || group_out == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_create: bad arguments");
  *group_out = nullptr;
  bool twice = false;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j) twice = twice || devs[i] == devs[j];
  if (comms != nullptr) {
    if (twice) return mppi_fail_message(MPPI_E_UNSUPPORTED, "mppi_group_create: RCCL takes one rank per device (a device is listed twice): pass comms = NULL (staged)");
    for (int i = 0; i < ndev; ++i)
      if (comms[i] == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_create: null communicator");
  }
  Group* g = new Group();
  g->n = ndev;
  g->staged = comms == nullptr;
  for (int i = 0; i < ndev; ++i) {
    Slot* s = new Slot();
    s->group = g; s->index = i; s->dev = devs[i]; s->comm = comms ? comms[i] : nullptr;
    s->submitted.store(1);
    g->slots.push_back(s);
  }
  for (Slot* s : g->slots) s->th = std::thread(worker, s);
  int rc = 0;
  for (Slot* s : g->slots) {
    while (s->done.load(std::memory_order_acquire) != 1) std::this_thread::yield();
    if (s->init_rc != 0 && rc == 0) rc = s->init_rc;
  }
  if (rc != 0) {
    mppi_group_destroy(g);
    return mppi_fail_message(rc, "mppi_group_create: a worker could not set its device / create its events");
  }
  const char* knob = getenv("MPPI_GROUP_IN_PLACE");               // "0": gather the records by copies instead (A/B, tests)
  g->in_place = g->staged && !(knob && knob[0] == '0');
  for (Slot* s : g->slots) g->in_place = g->in_place && s->peers_ok;
  *group_out = g;
  return 0;
}

extern "C" int mppi_group_destroy(void* group) {
  if (group == nullptr) return 0;
  Group* g = (Group*)group;
  if (g->pending != 0) (void)finish(g, /*abort=*/true, nullptr, nullptr);
  g->stop.store(true, std::memory_order_release);
  { std::lock_guard<std::mutex> lk(g->mu); g->cv.notify_all(); }
  for (Slot* s : g->slots) { if (s->th.joinable()) s->th.join(); delete s; }
  delete g;
  return 0;
}

extern "C" int mppi_group_size(void* group) { return group ? ((Group*)group)->n : 0; }

extern "C" int mppi_group_broadcast(void* group, const void* src, int64_t nbytes, void* const* dst, void* stream0) {
  Group* g = (Group*)group;
  if (g == nullptr || src == nullptr || dst == nullptr || nbytes <= 0) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_broadcast: bad arguments");
  if (g->pending != 0) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_broadcast: call it before the first mppi_group_submit of a command");
  bool any = false;
  for (int i = 0; i < g->n; ++i) {
    Slot* s = g->slots[i];
    s->bc_src = src; s->bc_dst = dst[i]; s->bc_bytes = (dst[i] != nullptr && dst[i] != src) ? nbytes : 0;
    any = any || s->bc_bytes > 0;
  }
  if (any) {
    const hipError_t e = hipEventRecord(g->bc_ev, (hipStream_t)stream0);
    if (e != hipSuccess) return mppi_fail_message((int)e, hipGetErrorString(e));
  }
  g->bc_armed = true;
  return 0;
}

extern "C" int mppi_group_submit(void* group, int32_t index, const MppiProblem* p, const MppiProblem* theta_problem, void* records,
                                 void* stream) {
  Group* g = (Group*)group;
  if (g == nullptr || index < 0 || index >= g->n || p == nullptr || records == nullptr)
    return mppi_fail_message(MPPI_E_BADARG, "mppi_group_submit: bad arguments");
  Slot* s = g->slots[index];
  if (s->cmd == g->cmd) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_submit: this device already holds its part of the command");
  const MppiProblem* q = theta_problem ? theta_problem : p;
  if (q->record == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_submit: the problem needs a record");
  if (g->pending == 0) {
    g->failed.store(0, std::memory_order_release);
    if (!g->bc_armed)
      for (Slot* o : g->slots) o->bc_bytes = 0;
    g->bc_armed = false;
  }
  s->p = *p;
  s->has_pt = theta_problem != nullptr;
  if (s->has_pt) s->pt = *theta_problem;
  s->records = records;
  s->stream = (hipStream_t)stream;
  s->cmd = g->cmd;
  ++g->pending;
  s->submitted.fetch_add(1, std::memory_order_seq_cst);
  if (s->sleeping.load(std::memory_order_seq_cst)) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->cv.notify_all();
  }
  return 0;
}

extern "C" int mppi_group_wait(void* group, int32_t* forms, int32_t* next_draws) {
  Group* g = (Group*)group;
  if (g == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_wait: null group");
  if (g->pending == 0) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_wait: nothing submitted");
  if (g->pending != g->n) {
    // a part is missing: the workers that did start must not enter the exchange (the collective would wait for it on the GPU)
    (void)finish(g, /*abort=*/true, forms, next_draws);
    return mppi_fail_message(MPPI_E_BADARG, "mppi_group_wait: every device of the group needs its part of the command (mppi_group_submit); the command was abandoned");
  }
  return finish(g, /*abort=*/false, forms, next_draws);
}

extern "C" int mppi_group_abort(void* group) {
  Group* g = (Group*)group;
  if (g == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_group_abort: null group");
  if (g->pending == 0) { g->bc_armed = false; return 0; }
  (void)finish(g, /*abort=*/true, nullptr, nullptr);
  return 0;
}
