"""One Python process, N devices: `MPPI(..., devices=[0, 1, ..., 7])`  (SURVEY.md 8b / 8e).

The reference's caller is ONE process stepping ONE environment (/root/reference/src/pytorch_mppi/mppi.py:876-898,
tests/pendulum.py:68-79): with `shard=(rank, world)` it has to re-launch its whole loop under torch.distributed.run and step a
replica of the environment on every rank.  `devices=[...]` keeps `.command(state)` the single drop-in call: the object the
constructor returns holds one shard controller per listed device (the same code a rank of the per-process model runs -- contiguous
split of the K samples by global index, K1 / K3 / K4 against the shard's own minimum, one (2 + T nu)-element record per shard), and
a command is

    state -> every device (peer copies)          U, the parameters and the model are replicated
    per device: K1, K3, K4(record only)          issued device by device from this thread, each on its device's current stream
    ONE exchange                                  the N all-gathers of the records inside one ncclGroupStart / ncclGroupEnd
                                                  (C-ABI mppi_exchange_combine_all, communicators from ncclCommInitAll), then
    per device: K5                                the rank-order combine -> bit-identical U on every device
    return device 0's action

A device listed more than once (`devices=[0, 0]`: the one-GPU test rig) or a process without RCCL stages the records through
device copies instead of RCCL (which takes one rank per device); everything else is the same code.

The returned object is an instance of the class that was asked for (a subclass made on the fly), but holds no controller state of
its own: attribute reads go to shard 0 -- except the per-sample results (`cost_total`, `omega`, `noise`, ...), which are the
shards' parts concatenated on device 0 in global sample order --, attribute writes go to every shard (tensors moved to the shard's
device), methods other than `command` run on every shard and the replicated sequences (`U`, `theta`, `action_sequence`) are then
re-copied from shard 0 (a `reset()` draws on device 0 only, like rank 0's draw is broadcast in the per-process model).

Host cost: the shards' launches are issued one after the other by one Python thread (~40 us each), so commands shorter than
N x 40 us are host-bound here; the per-process model (`shard=`, what `bench.py` runs under torch.distributed.run) has no such
limit.  For C5-sized commands (0.5 ms) it does not matter."""
import contextlib
import ctypes as C

import torch

from . import _native as N

# per-sample results of the last command: name -> the axis that is the sample axis
_PER_SAMPLE = {"cost_total": 0, "omega": 0, "cost_total_non_zero": 0, "noise": 0, "perturbed_action": 0, "noise_theta": 0,
               "perturbed_control": 0, "states": 1, "actions": 1}
_REPLICATED = ("U", "theta", "action_sequence")
_OWN = frozenset(("_shards", "_devs", "_comms", "_staged", "_base", "exchange"))
_classes = {}


def group_class(cls):
    """the class of the object `cls(..., devices=[d0, d1, ...])` returns: (DeviceGroup, cls)"""
    g = _classes.get(cls)
    if g is None:
        g = _classes[cls] = type(cls.__name__ + "OnDevices", (DeviceGroup, cls), {"_base": cls})
    return g


def _on(device):
    """the shard's device as the calling thread's current one (what the engine's launches and torch's allocations follow)"""
    return torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()


def _dev_index(d):
    if isinstance(d, int):
        return d
    d = torch.device(d)
    if d.type != "cuda":
        raise ValueError(f"devices= takes GPU indices / cuda devices, not {d}")
    return d.index if d.index is not None else torch.cuda.current_device()


class DeviceGroup:
    def __init__(self, *args, devices=None, **kw):
        cls = type(self)._base
        devs = [_dev_index(d) for d in devices]
        if len(devs) < 2:
            raise ValueError("a device group needs at least two entries in devices=")
        kw.pop("shard", None)
        kw.pop("device", None)
        args = list(args)
        shards = []
        for g, dv in enumerate(devs):
            a = list(args)
            k = dict(kw)
            if len(a) > 6:
                a[6] = torch.device("cuda", dv)          # `device` given positionally (mppi.py:45-61 order)
            else:
                k["device"] = torch.device("cuda", dv)
            with torch.cuda.device(dv):
                shards.append(cls(*a, shard=(g, len(devs)), **k))
        object.__setattr__(self, "_shards", shards)
        object.__setattr__(self, "_devs", devs)
        object.__setattr__(self, "_comms", None)
        object.__setattr__(self, "_staged", len(set(devs)) < len(devs))
        object.__setattr__(self, "exchange", None)
        self._sync_replicated()
        if not self._staged:
            lib = N.lib()
            comms = (C.c_void_p * len(devs))()
            rc = lib.mppi_dist_init_all(len(devs), (C.c_int32 * len(devs))(*devs), comms)
            if rc == 0:
                object.__setattr__(self, "_comms", comms)
                import weakref
                weakref.finalize(self, DeviceGroup._destroy, lib, [C.c_void_p(c) for c in comms])
            elif rc != N.E_UNSUPPORTED:
                N.check(rc, "mppi_dist_init_all")
            else:
                object.__setattr__(self, "_staged", True)          # no RCCL in this process: device copies
        object.__setattr__(self, "exchange", "staged through device copies" + (" (a device is listed twice: TEST RIG)" if len(set(devs)) < len(devs) else
                                                                               " (no RCCL)") if self._staged else
                           "engine-owned RCCL communicators (ncclCommInitAll), one grouped all-gather per command")

    @staticmethod
    def _destroy(lib, comms):
        for c in comms:
            try:
                if c.value:
                    lib.mppi_dist_destroy(c)
            except Exception:
                pass

    # ---- attribute plumbing --------------------------------------------------------------------------------------------
    def __getattribute__(self, name):
        if name in _OWN or name.startswith("__") or name in DeviceGroup.__dict__:
            return object.__getattribute__(self, name)
        shards = object.__getattribute__(self, "_shards")
        if name in _PER_SAMPLE:
            parts = [getattr(s, name) for s in shards]
            if any(p_ is None for p_ in parts):
                return None
            d0 = shards[0].d
            return torch.cat([p_.to(d0) for p_ in parts], dim=_PER_SAMPLE[name])
        v = getattr(shards[0], name)
        if callable(v) and getattr(v, "__self__", None) is shards[0]:
            def on_every_shard(*a, **k):
                out = None
                for i, s in enumerate(shards):
                    with _on(s.d):
                        r = getattr(s, name)(*[self._to(x, s) for x in a], **{kk: self._to(x, s) for kk, x in k.items()})
                    if i == 0:
                        out = r
                self._sync_replicated()
                return out
            return on_every_shard
        return v

    def __setattr__(self, name, value):
        for s in object.__getattribute__(self, "_shards"):
            setattr(s, name, self._to(value, s))

    @staticmethod
    def _to(v, shard):
        if torch.is_tensor(v) and v.is_cuda and v.device != shard.d:
            return v.to(shard.d)
        return v

    def _sync_replicated(self):
        """the sequences every shard must hold identically are shard 0's (a constructor / reset() draw happens per device)"""
        shards = object.__getattribute__(self, "_shards")
        s0 = shards[0]
        for name in _REPLICATED:
            v = s0.__dict__.get(name)
            if torch.is_tensor(v):
                for s in shards[1:]:
                    s.__dict__[name] = v.to(s.d, copy=True)

    @property
    def devices(self):
        return list(object.__getattribute__(self, "_devs"))

    @property
    def shards(self):
        return list(object.__getattribute__(self, "_shards"))

    # ---- one command ---------------------------------------------------------------------------------------------------
    def command(self, state, shift_nominal_trajectory=True, info=None):
        """mppi.py:240-252 on N devices: the action (device 0), without synchronising."""
        shards = object.__getattribute__(self, "_shards")
        s0 = shards[0]
        ps = []
        for s in shards:
            s.info = info
            with _on(s.d):
                if s._jit_pending is not None:
                    s._adopt_background_model()
                if getattr(s._model, "watch", None) is not None:
                    s._check_traced(state if s is s0 else None)
                # (MPPI._to_state moves the state to the shard's device and, for per-sample initial states of the global
                # problem -- (K, nx), mppi.py:302-305 --, takes this shard's rows)
                ps.append(s._begin(state, bool(shift_nominal_trajectory)))
        self._exchange(ps)
        action = None
        for s, p in zip(shards, ps):
            with _on(s.d):
                a = s._end(p)
            if s is s0:
                action = a
        return action

    def _exchange(self, ps):
        shards = object.__getattribute__(self, "_shards")
        G = len(shards)
        recs = [p._keep["record"] for p in ps]
        comms = object.__getattribute__(self, "_comms")
        if comms is not None:
            n = recs[0].numel()
            bufs = []
            for s, p in zip(shards, ps):
                b = torch.empty(G, n, device=s.d, dtype=s.dtype)
                p._keep["records"] = b
                bufs.append(b)
            devs = object.__getattribute__(self, "_devs")
            streams = [torch.cuda.current_stream(s.d).cuda_stream for s in shards]
            N.check(N.lib().mppi_exchange_combine_all(
                G, (C.c_int32 * G)(*devs), (C.POINTER(N.MppiProblem) * G)(*[C.pointer(p) for p in ps]), comms,
                (C.c_void_p * G)(*[b.data_ptr() for b in bufs]), (C.c_void_p * G)(*streams)), "mppi_exchange_combine_all")
            return
        # staged: the records travel by device copies (torch orders them against the streams involved), K5 on every device
        for s, p in zip(shards, ps):
            with _on(s.d):
                s._combine(p, torch.stack([r if r.device == s.d else r.to(s.d) for r in recs]))
