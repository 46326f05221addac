// dist.hip -- the multi-GPU exchange of a sharded command inside the C-ABI.
//
// The reference has no multi-GPU path.  Here the sample axis K is sharded over one process per GPU;
// the only data-path collective of a command is ONE all-gather of the (2 + J)-element shard record
// {beta_g, eta_g, P_g[J]} (3.1 KB at C3), after which every rank runs the same rank-order combine
// (K5).  Issuing that collective through torch.distributed costs ~28 us of host time per command and
// two cross-stream waits (its kernel runs on a pool stream); here RCCL's own C API is called on the
// CALLER'S stream, so a sharded command is K1, K3, K4, ncclAllGather, K5 back to back on one stream,
// from one C call (mppi_command_sharded).
//
// RCCL is bound at run time (dlsym on the already-loaded image first -- torch ships and loads its own
// librccl -- then dlopen): the library has no link-time dependency on it, loads on machines without
// RCCL, and never puts a second RCCL instance beside torch's.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include "common.hpp"

namespace {
struct NcclUniqueId { char internal[128]; };                     // rccl.h:43 (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_comm_destroy)(NcclComm);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef int (*fn_comm_init_all)(NcclComm*, int, const int*);
typedef int (*fn_group)(void);
typedef const char* (*fn_error_string)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8;                // ncclDataType_t

struct Rccl {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_comm_init_all comm_init_all = nullptr;      // single process, N devices (mppi_dist_init_all); optional
  fn_group group_start = nullptr, group_end = nullptr;
  fn_error_string error_string = nullptr;
  bool tried = false, ok = false;
  char why[200] = "";
};
Rccl g_rccl;

void* find_symbol(void** handle, const char* name) {
  if (void* s = dlsym(RTLD_DEFAULT, name)) return s;              // torch's librccl, if the process holds one
  if (*handle == nullptr) {
    const char* env = getenv("MPPI_RCCL_LIB");
    const char* cands[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* c : cands) {
      if (c == nullptr || c[0] == 0) continue;
      *handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
      if (*handle != nullptr) break;
    }
  }
  return *handle != nullptr ? dlsym(*handle, name) : nullptr;
}

bool rccl_load() {
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  void* h = nullptr;
  g_rccl.get_unique_id = (fn_get_unique_id)find_symbol(&h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)find_symbol(&h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)find_symbol(&h, "ncclCommDestroy");
  g_rccl.all_gather = (fn_all_gather)find_symbol(&h, "ncclAllGather");
  g_rccl.error_string = (fn_error_string)find_symbol(&h, "ncclGetErrorString");
  g_rccl.comm_init_all = (fn_comm_init_all)find_symbol(&h, "ncclCommInitAll");
  g_rccl.group_start = (fn_group)find_symbol(&h, "ncclGroupStart");
  g_rccl.group_end = (fn_group)find_symbol(&h, "ncclGroupEnd");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_gather;
  if (!g_rccl.ok) snprintf(g_rccl.why, sizeof(g_rccl.why), "RCCL not found (set MPPI_RCCL_LIB): %s", dlerror() ? dlerror() : "symbols missing");
  return g_rccl.ok;
}
}  // namespace

// set by capi.hip
int mppi_fail_message(int code, const char* msg);

extern "C" int mppi_dist_available(void) { return rccl_load() ? 1 : 0; }

extern "C" int mppi_dist_unique_id(void* id128) {
  if (id128 == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_dist_unique_id: null buffer");
  if (!rccl_load()) return mppi_fail_message(MPPI_E_UNSUPPORTED, g_rccl.why);
  NcclUniqueId id;
  const int r = g_rccl.get_unique_id(&id);
  if (r != 0) return mppi_fail_message(MPPI_E_DIST, g_rccl.error_string ? g_rccl.error_string(r) : "ncclGetUniqueId failed");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int mppi_dist_init(const void* id128, int32_t rank, int32_t world_size, void** comm_out) {
  if (id128 == nullptr || comm_out == nullptr || world_size <= 0 || rank < 0 || rank >= world_size)
    return mppi_fail_message(MPPI_E_BADARG, "mppi_dist_init: bad arguments");
  if (!rccl_load()) return mppi_fail_message(MPPI_E_UNSUPPORTED, g_rccl.why);
  NcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NcclComm comm = nullptr;
  const int r = g_rccl.comm_init_rank(&comm, world_size, id, rank);
  if (r != 0) return mppi_fail_message(MPPI_E_DIST, g_rccl.error_string ? g_rccl.error_string(r) : "ncclCommInitRank failed");
  *comm_out = comm;
  return 0;
}

extern "C" int mppi_dist_destroy(void* comm) {
  if (comm == nullptr) return 0;
  if (!rccl_load()) return mppi_fail_message(MPPI_E_UNSUPPORTED, g_rccl.why);
  const int r = g_rccl.comm_destroy((NcclComm)comm);
  return r == 0 ? 0 : mppi_fail_message(MPPI_E_DIST, "ncclCommDestroy failed");
}

// the ONE data-path collective of a sharded command: p->record -> records[world][2 + J] (also called by the device group's
// workers, csrc/group.hip: one thread per device, each on its own communicator)
namespace mppi_dist {
int all_gather_record(const MppiProblem* p, void* comm, void* records, void* stream) {
  if (p == nullptr || comm == nullptr || records == nullptr || p->record == nullptr)
    return mppi_fail_message(MPPI_E_BADARG, "the record all-gather needs a problem with a record, a communicator and the records buffer");
  if (!rccl_load()) return mppi_fail_message(MPPI_E_UNSUPPORTED, g_rccl.why);
  const size_t n = 2 + (size_t)p->T * p->nu;                       // elements of one shard record
  const int dt = p->dtype == MPPI_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32;
  const int r = g_rccl.all_gather(p->record, records, n, dt, (NcclComm)comm, (hipStream_t)stream);
  if (r != 0) return mppi_fail_message(MPPI_E_DIST, g_rccl.error_string ? g_rccl.error_string(r) : "ncclAllGather failed");
  return 0;
}
}  // namespace mppi_dist

extern "C" int mppi_exchange_combine(const MppiProblem* p, void* comm, void* records, int32_t world_size, void* stream) {
  if (world_size <= 0) return mppi_fail_message(MPPI_E_BADARG, "mppi_exchange_combine: world_size <= 0");
  if (int e = mppi_dist::all_gather_record(p, comm, records, stream)) return e;
  return mppi_combine(p, records, world_size, stream);
}

extern "C" int mppi_command_sharded(const MppiProblem* p, void* comm, void* records, int32_t world_size, void* stream) {
  if (int e = mppi_command(p, /*apply=*/0, stream)) return e;      // K1, K3, K4: this shard's record
  MppiProblem q = *p;
  if (q.noise_src == MPPI_NOISE_PHILOX && q.z != nullptr) q.noise_src = MPPI_NOISE_TNK4;
  return mppi_exchange_combine(&q, comm, records, world_size, stream);
}

// ---- one process, N devices ("single Python process, ncclCommInitAll over visible devices, one stream per
// device, ncclGroupStart/End -- keeps .command() a single drop-in call") ----
extern "C" int mppi_dist_init_all(int32_t ndev, const int32_t* devs, void** comms_out) {
  if (ndev <= 0 || devs == nullptr || comms_out == nullptr) return mppi_fail_message(MPPI_E_BADARG, "mppi_dist_init_all: bad arguments");
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (devs[i] == devs[j]) return mppi_fail_message(MPPI_E_UNSUPPORTED, "mppi_dist_init_all: RCCL takes one rank per device (a device is listed twice)");
  if (!rccl_load()) return mppi_fail_message(MPPI_E_UNSUPPORTED, g_rccl.why);
  if (!g_rccl.comm_init_all || !g_rccl.group_start || !g_rccl.group_end)
    return mppi_fail_message(MPPI_E_UNSUPPORTED, "this RCCL has no ncclCommInitAll / ncclGroupStart / ncclGroupEnd");
  const int r = g_rccl.comm_init_all((NcclComm*)comms_out, ndev, (const int*)devs);
  if (r != 0) return mppi_fail_message(MPPI_E_DIST, g_rccl.error_string ? g_rccl.error_string(r) : "ncclCommInitAll failed");
  return 0;
}

// The exchange of ONE command on all devices of the process: the N all-gathers of the shard records inside one RCCL group
// (problems[g]->record -> records[g][ndev][2 + J], on streams[g], the stream device devs[g]'s part of the command was issued
// on), then K5 on every device.  The calling thread's current device is restored.
extern "C" int mppi_exchange_combine_all(int32_t ndev, const int32_t* devs, const MppiProblem* const* problems, void* const* comms,
                                         void* const* records, void* const* streams) {
  if (ndev <= 0 || devs == nullptr || problems == nullptr || comms == nullptr || records == nullptr || streams == nullptr)
    return mppi_fail_message(MPPI_E_BADARG, "mppi_exchange_combine_all: bad arguments");
  if (!rccl_load() || !g_rccl.group_start || !g_rccl.group_end) return mppi_fail_message(MPPI_E_UNSUPPORTED, g_rccl.why);
  for (int g = 0; g < ndev; ++g)
    if (problems[g] == nullptr || problems[g]->record == nullptr || comms[g] == nullptr || records[g] == nullptr)
      return mppi_fail_message(MPPI_E_BADARG, "mppi_exchange_combine_all: every device needs a problem with a record, a communicator and a records buffer");
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  const size_t n = 2 + (size_t)problems[0]->T * problems[0]->nu;
  const int dt = problems[0]->dtype == MPPI_F64 ? NCCL_FLOAT64 : NCCL_FLOAT32;
  int r = g_rccl.group_start();
  for (int g = 0; g < ndev && r == 0; ++g) {
    (void)hipSetDevice(devs[g]);
    r = g_rccl.all_gather(problems[g]->record, records[g], n, dt, (NcclComm)comms[g], (hipStream_t)streams[g]);
  }
  const int r2 = g_rccl.group_end();
  if (r == 0) r = r2;
  int e = 0;
  if (r != 0) e = mppi_fail_message(MPPI_E_DIST, g_rccl.error_string ? g_rccl.error_string(r) : "ncclAllGather (group) failed");
  for (int g = 0; g < ndev && e == 0; ++g) {
    (void)hipSetDevice(devs[g]);
    e = mppi_combine(problems[g], records[g], ndev, streams[g]);
  }
  (void)hipSetDevice(dev0);
  return e;
}
