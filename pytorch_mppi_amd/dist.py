"""Multi-GPU sharding of the sample axis K (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).

The reference has no multi-GPU path.  K is split contiguously by global sample
index -- rank g owns [k_offset, k_offset + K_local) -- so the row bookkeeping of
`sample_null_action` / sampler rows (mppi.py:387-400) and the Philox counters are independent of
the number of ranks.  U, the state and all parameters are replicated.  Each rank reduces its
shard against its OWN minimum beta_g (K3/K4) into a record {beta_g, eta_g, P_g[T*nu]}; ONE
all-gather of that (T*nu+2)-element record is the only collective of a command; every rank
then combines the records in rank order (mppi_combine, K5), which yields bit-identical U on
all ranks:   beta = min beta_g,  s_g = exp(-(beta_g-beta)/lambda),
             eta = sum_g s_g eta_g,  U += sum_g s_g P_g / eta.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


class NativeComm:
    """An RCCL communicator owned by the C-ABI (include/mppi_amd.h, csrc/dist.hip): the record
    all-gather of a sharded command is then issued by the engine itself on the caller's stream --
    no torch.distributed call, no pool stream, no cross-stream wait on the per-command path.
    torch.distributed (any backend) is only used at start-up: to agree that EVERY rank can take this path
    (a rank that silently fell back to torch.distributed while the others call ncclAllGather would hang the
    job), and to ship rank 0's 128-byte RCCL id.  Destroyed by a finalizer (or `close()`)."""

    def __init__(self, rank, world_size, device, group=None):
        import weakref
        from . import _native as N
        lib = N.lib()
        multi = world_size > 1
        if multi and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("NativeComm with world_size > 1 needs an initialised torch.distributed group for the "
                    "id hand-out")
        on_dev = multi and dist.get_backend(group) == "nccl"
        flag_dev = device if on_dev else "cpu"

        def all_ranks(ok):
            """collective AND over the ranks of the group (identity at world_size 1)"""
            if not multi:
                return bool(ok)
            f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
            return bool(int(f.item()))

        # 1. everything that can fail locally BEFORE any rank enters the id broadcast: library support, and on
        #    rank 0 the id itself
        idbuf = (C.c_char * 128)()
        why = ""
        ok = bool(lib.mppi_dist_available())
        if not ok:
            why = "RCCL not found by the engine library: " + lib.mppi_last_error().decode(errors="replace")
        if ok and rank == 0 and lib.mppi_dist_unique_id(idbuf) != 0:
            ok, why = False, "mppi_dist_unique_id: " + lib.mppi_last_error().decode(errors="replace")
        if not all_ranks(ok):
            raise RuntimeError(why or "another rank cannot use the engine-owned RCCL communicator")
        # 2. ship the id
        if multi:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            t = torch.zeros(128, dtype=torch.uint8, device=flag_dev)
            if rank == 0:
                t.copy_(torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8))
            dist.broadcast(t, src=src, group=group)
            idbuf.raw = bytes(t.cpu().numpy().tobytes())
        # 3. ncclCommInitRank on every rank, then agree on the outcome: one failure sends ALL ranks back to
        #    torch.distributed together
        comm = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.mppi_dist_init(idbuf, int(rank), int(world_size), C.byref(comm))
        why = "" if rc == 0 else "mppi_dist_init: " + lib.mppi_last_error().decode(errors="replace")
        if not all_ranks(rc == 0):
            if rc == 0 and comm.value:
                lib.mppi_dist_destroy(comm)
            raise RuntimeError(why or "mppi_dist_init failed on another rank")
        self._lib, self.handle, self.world_size = lib, comm, int(world_size)
        self._finalizer = weakref.finalize(self, NativeComm._destroy, lib, comm)

    @staticmethod
    def _destroy(lib, handle):
        if handle is not None and handle.value:
            try:
                lib.mppi_dist_destroy(handle)
            except Exception:       # interpreter shutdown: the library may be gone already
                pass

    def close(self):
        if self.handle is not None:
            self._finalizer.detach()
            NativeComm._destroy(self._lib, self.handle)
            self.handle = None


# ShardPlan(..., group=LOCAL): the shards are the devices of ONE process (group.py), not ranks of a process group
LOCAL = "local"


class ShardPlan:
    def __init__(self, K, rank, world_size, group=None):
        if not (0 <= rank < world_size):
            raise ValueError("bad rank/world_size")
        if K < world_size:
            raise ValueError("need at least one sample per rank")
        # a process-local plan (a device group's shard: `rank` is the shard's index, not a process rank) never touches
        # torch.distributed -- in a process whose default group is RCCL-backed (a device group per torchrun rank) a
        # broadcast of U or an ncclCommInitRank with shard indices for ranks would hang or mix the ranks' sequences
        # 
        self.local = isinstance(group, str) and group == LOCAL
        if self.local:
            group = None
        self.K, self.rank, self.world_size, self.group = int(K), int(rank), int(world_size), group
        base, rem = divmod(self.K, self.world_size)
        self.K_local = base + (1 if rank < rem else 0)
        self.k_offset = rank * base + min(rank, rem)
        self._native = None            # NativeComm | False (tried, unavailable) | None (not tried)

    def native_comm(self, device):
        """The engine-owned RCCL communicator for this plan, or None: when the process group is
        RCCL-backed (backend "nccl": one rank per GPU), or at world_size 1 (measurement / test rig).
        MPPI_NATIVE_RCCL=0 keeps the exchange on torch.distributed.  The decision is COLLECTIVE: whether this
        rank wants the native path (environment, device) enters the same all-reduce(MIN) as what can fail
        inside NativeComm, so either every rank gets a communicator or every rank gets None."""
        if self.local:
            return None                 # the device group owns the exchange (csrc/group.hip)
        if self._native is None:
            self._native = False
            group_ok = self.world_size == 1 or (dist.is_available() and dist.is_initialized()
                                                and dist.get_backend(self.group) == "nccl")
            if group_ok:        # same on every rank (a property of the group), so every rank enters or none does
                want = os.environ.get("MPPI_NATIVE_RCCL", "1") != "0" and torch.device(device).type == "cuda"
                if self.world_size > 1:
                    f = torch.tensor([1 if want else 0], dtype=torch.int32, device=device)
                    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=self.group)
                    want = bool(int(f.item()))
                if want:
                    try:
                        self._native = NativeComm(self.rank, self.world_size, device, self.group)
                    except RuntimeError:
                        self._native = False
        return self._native or None

    def bounds(self, rank):
        base, rem = divmod(self.K, self.world_size)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)

    def all_gather(self, record):
        """(2+J,) shard record -> (world_size, 2+J), rank order.  The single collective."""
        n = record.numel()
        if record.is_cuda and dist.get_backend(self.group) == "gloo":
            # test rigs only (several ranks sharing ONE GPU, where RCCL refuses to run): stage the
            # record through the host.  Production = nccl (RCCL), device to device.
            host = torch.empty(self.world_size * n, dtype=record.dtype)
            dist.all_gather_into_tensor(host, record.detach().cpu().contiguous().view(-1), group=self.group)
            return host.to(record.device).view(self.world_size, n)
        out = torch.empty(self.world_size * n, device=record.device, dtype=record.dtype)
        dist.all_gather_into_tensor(out, record.contiguous().view(-1), group=self.group)
        return out.view(self.world_size, n)


    def all_gather_start(self, record):
        """Same collective, not waited for: returns (out, work).  `work.wait()` makes the caller's
        stream wait for it; until then the caller may queue work that does not depend on the records
        (the next command's noise generation) and the latency-bound collective hides behind it."""
        if not record.is_cuda or dist.get_backend(self.group) == "gloo":
            return self.all_gather(record), None          # test rigs: host-staged, synchronous
        n = record.numel()
        out = torch.empty(self.world_size * n, device=record.device, dtype=record.dtype)
        work = dist.all_gather_into_tensor(out, record.contiguous().view(-1), group=self.group, async_op=True)
        return out.view(self.world_size, n), work


def combine_records_host(records, U_eff, lambda_):
    """Host restatement of mppi_combine (K5) in torch ops -- used by the CPU (gloo) tests to check
    the exchange + rank-order combine logic, and by the GPU tests as the checker of the kernel.
    records: (G, 2+J); U_eff: (T,nu) nominal sequence after the shift.  Returns (U_new, beta, eta)."""
    beta_g = records[:, 0]
    beta = beta_g.min()
    s = torch.exp(-(1.0 / lambda_) * (beta_g - beta))
    eta = torch.zeros((), dtype=records.dtype, device=records.device)
    P = torch.zeros(records.shape[1] - 2, dtype=records.dtype, device=records.device)
    for g in range(records.shape[0]):          # fixed rank order, like the kernel
        eta = eta + s[g] * records[g, 1]
        P = P + s[g] * records[g, 2:]
    U_new = U_eff + (P * (1.0 / eta)).reshape(U_eff.shape)
    return U_new, beta, eta
