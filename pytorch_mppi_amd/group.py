"""One Python process, N devices: `MPPI(..., devices=[0, 1, ..., 7])`  (SURVEY.md 8b / 8e).

The reference's caller is ONE process stepping ONE environment (/root/reference/src/pytorch_mppi/mppi.py:876-898,
tests/pendulum.py:68-79): with `shard=(rank, world)` it has to re-launch its whole loop under torch.distributed.run and
step a replica of the environment on every rank.  `devices=[...]` keeps `.command(state)` the single drop-in call: the
object the constructor returns holds one shard controller per listed device (the same code a rank of the per-process
model runs -- contiguous split of the K samples by global index, K1 / K3 / K4 against the shard's own minimum, one (2 +
T nu)-element record per shard), and a command is

    this thread:  the state to device 0; per shard the problem block of this command (`MPPI._prepare`: draw, buffers,
    parameters)
                  -> handed to the ENGINE's device group (C-ABI 22, csrc/group.hip: mppi_group_broadcast / _submit /
                  _wait)
    worker g:     one thread per device inside the library, that device current in it: the state from device 0 (peer
    copy),
                  K1, K3, K4 (record only), the exchange -- ncclAllGather on the device's own communicator
                  (ncclCommInitAll), or copies behind events when a device is listed twice / there is no RCCL ("staged")
                  -- and K5, the rank-order
                  combine -> bit-identical U on every device
    this thread:  return device 0's action

so the host's share of a command is N block hand-overs plus ONE device's launches, not N x (Python + launches): commands
no longer have to be longer than N x 40 us to be GPU-bound (profiles/r06_group_host_issue.txt).  What has no one-call
form (the callback path, KMPPI's two-launch form) is issued shard by shard from this thread as before, the exchange
through mppi_exchange_combine_all / device copies.  MPPI_GROUP_THREADS=0 forces that serial form (A/B).

The returned object is an instance of the class that was asked for (a subclass made on the fly), but holds no controller
state of its own: attribute reads go to shard 0 -- except the per-sample results (`cost_total`, `omega`, `noise`, ...),
which are the shards' parts concatenated on device 0 in global sample order --, attribute writes go to every shard
(tensors moved to the shard's device), methods other than `command` run on every shard and the replicated sequences
(`U`, `theta`, `action_sequence`) are then
re-copied from shard 0 (a `reset()` draws on device 0 only, like rank 0's draw is broadcast in the per-process model).

User callables (the reference's plugin API) are called by every shard with tensors on THAT shard's device.
`torch.nn.Module`s --
the callable itself or the object a bound method belongs to: /root/reference/tests/pendulum_approximate.py's network --
are deep-copied onto each further device at construction and re-synchronised (`load_state_dict`) whenever the original's
parameters were written (retraining between commands, mppi.py:890-893); `models.NativeModel`s keep a parameter blob per
device themselves. Anything else -- a closure over a tensor on cuda:0 -- must be device-agnostic
(`tensor.to(state.device)`): the error a device
mismatch raises inside such a callable is re-raised with that advice."""
import contextlib
import ctypes as C
import os
import time

import torch

from . import _native as N

# per-sample results of the last command: name -> the axis that is the sample axis
_PER_SAMPLE = {"cost_total": 0, "omega": 0, "cost_total_non_zero": 0, "noise": 0, "perturbed_action": 0,
        "noise_theta": 0,
               "perturbed_control": 0, "states": 1, "actions": 1}
_REPLICATED = ("U", "theta", "action_sequence")
_classes = {}


def group_class(cls):
    """the class of the object `cls(..., devices=[d0, d1, ...])` returns: (DeviceGroup, cls)"""
    g = _classes.get(cls)
    if g is None:
        g = _classes[cls] = type(cls.__name__ + "OnDevices", (DeviceGroup, cls), {"_base": cls})
    return g


def _on(device):
    """the shard's device as the calling thread's current one (what the engine's launches and torch's allocations
    follow)"""
    return torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()


def _dev_index(d):
    if isinstance(d, int):
        return d
    d = torch.device(d)
    if d.type != "cuda":
        raise ValueError(f"devices= takes GPU indices / cuda devices, not {d}")
    return d.index if d.index is not None else torch.cuda.current_device()


_OWN = frozenset(("_shards", "_devs", "_comms", "_staged", "_base", "exchange", "_engine", "_replicas", "_state_bufs",
        "_rec_bufs",
                  "_threads", "issue", "_fast", "_dirty", "wait_seconds", "_rearm"))


def _module_of(fn):
    """the torch.nn.Module a user callable is (or is a bound method of), else None"""
    if isinstance(fn, torch.nn.Module):
        return fn, None
    owner = getattr(fn, "__self__", None)
    if isinstance(owner, torch.nn.Module) and getattr(fn, "__name__", None):
        return owner, fn.__name__
    return None, None


class _Replicas:
    """nn.Modules among the user's callables, copied to the further devices of a group and kept equal to the
    originals"""

    def __init__(self):
        self.items = {}          # id(original) -> (original, {device: copy}, versions at the last sync)

    def on(self, fn, device):
        mod, name = _module_of(fn)
        if mod is None:
            return fn
        it = self.items.get(id(mod))
        if it is None:
            it = self.items[id(mod)] = [mod, {}, self._versions(mod)]
        cp = it[1].get(device)
        if cp is None:
            import copy
            cp = it[1][device] = copy.deepcopy(mod).to(device)
        return cp if name is None else getattr(cp, name)

    @staticmethod
    def _versions(mod):
        return tuple(t._version for t in list(mod.parameters()) + list(mod.buffers()))

    def sync(self):
        """before a command: originals written since the last look (an optimizer step, load_state_dict) -> copies
        follow"""
        for it in self.items.values():
            v = self._versions(it[0])
            if v != it[2]:
                sd = it[0].state_dict()
                for dev, cp in it[1].items():
                    cp.load_state_dict({k: t.to(dev) for k, t in sd.items()})
                it[2] = v


class DeviceGroup:
    def __init__(self, *args, devices=None, **kw):
        import os
        from .dist import LOCAL
        cls = type(self)._base
        devs = [_dev_index(d) for d in devices]
        if len(devs) < 2:
            raise ValueError("a device group needs at least two entries in devices=")
        kw.pop("shard", None)
        kw.pop("device", None)
        args = list(args)
        shards = []
        replicas = _Replicas()
        dev0 = torch.device("cuda", devs[0])
        for g, dv in enumerate(devs):
            a = list(args)
            k = dict(kw)
            d = torch.device("cuda", dv)
            if len(a) > 6:
                a[6] = d                                 # `device` given positionally (mppi.py:45-61 order)
            else:
                k["device"] = d
            if d != dev0:
                # the user's callables see tensors of THIS device: modules travel with the shard (see the module
                # docstring)
                for i in (0, 1):
                    if len(a) > i:
                        a[i] = replicas.on(a[i], d)
                for name in ("dynamics", "running_cost", "terminal_state_cost"):
                    if k.get(name) is not None:
                        k[name] = replicas.on(k[name], d)
                if len(a) > 7 and a[7] is not None:
                    a[7] = replicas.on(a[7], d)          # terminal_state_cost given positionally
            with torch.cuda.device(dv):
                shards.append(cls(*a, shard=(g, len(devs), LOCAL), **k))
        object.__setattr__(self, "_shards", shards)
        object.__setattr__(self, "_devs", devs)
        object.__setattr__(self, "_comms", None)
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_replicas", replicas)
        object.__setattr__(self, "_state_bufs", None)
        object.__setattr__(self, "_rec_bufs", {})
        object.__setattr__(self, "_fast", None)
        object.__setattr__(self, "_dirty", True)
        object.__setattr__(self, "_rearm", os.environ.get("MPPI_GROUP_REARM", "1") != "0")      # (A/B, tests: "0" = rebuild every block)
        # seconds the calling thread has spent in mppi_group_wait (the workers issuing their launches): the rest of a command's
        # issue time is the caller's own share -- what bench.py's host probe tells apart (on a rig whose shards share ONE device the
        # workers' launches serialise on that device's queue, which N real devices do not)
        object.__setattr__(self, "wait_seconds", 0.0)
        object.__setattr__(self, "_staged", len(set(devs)) < len(devs))
        object.__setattr__(self, "exchange", None)
        object.__setattr__(self, "issue", None)
        self._sync_replicated()
        lib = N.lib()
        import weakref
        if not self._staged:
            comms = (C.c_void_p * len(devs))()
            rc = lib.mppi_dist_init_all(len(devs), (C.c_int32 * len(devs))(*devs), comms)
            if rc == 0:
                object.__setattr__(self, "_comms", comms)
                weakref.finalize(self, DeviceGroup._destroy, lib, [C.c_void_p(c) for c in comms])
            elif rc != N.E_UNSUPPORTED:
                N.check(rc, "mppi_dist_init_all")
            else:
                object.__setattr__(self, "_staged", True)          # no RCCL in this process: device copies
        object.__setattr__(self, "exchange",
                "staged through device copies" + (" (a device is listed twice: TEST RIG)"
                        if len(set(devs)) < len(devs) else
                                                                               " (no RCCL)") if self._staged else
                           "engine-owned RCCL communicators (ncclCommInitAll), one all-gather per device and command")
        # the engine's device group: one worker thread per device issues that device's launches (csrc/group.hip)
        object.__setattr__(self, "_threads", os.environ.get("MPPI_GROUP_THREADS", "1") != "0")
        if self._threads:
            eng = C.c_void_p()
            rc = lib.mppi_group_create(len(devs), (C.c_int32 * len(devs))(*devs), self._comms, C.byref(eng))
            if rc == 0:
                object.__setattr__(self, "_engine", eng)
                weakref.finalize(self, DeviceGroup._destroy_engine, lib, eng)
            else:
                import logging
                logging.getLogger("pytorch_mppi_amd").warning(
                    "pytorch_mppi_amd: no engine device group (%s): the shards' launches are issued from the calling "
                            "thread",
                    lib.mppi_last_error().decode(errors="replace"))
        object.__setattr__(self, "issue", "one worker thread per device inside the engine (mppi_group_submit / "
                "mppi_group_wait)"
                           if self._engine is not None else "shard by shard from the calling thread")

    @staticmethod
    def _destroy_engine(lib, eng):
        try:
            if eng.value:
                lib.mppi_group_destroy(eng)
        except Exception:
            pass

    @staticmethod
    def _destroy(lib, comms):
        for c in comms:
            try:
                if c.value:
                    lib.mppi_dist_destroy(c)
            except Exception:
                pass

    # ---- attribute plumbing
    # --------------------------------------------------------------------------------------------
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
                object.__setattr__(self, "_dirty", True)
                out = None
                for i, s in enumerate(shards):
                    with _on(s.d):
                        r = getattr(s, name)(*[self._to(x, s) for x in a], **{kk: self._to(x, s) for kk,
                                x in k.items()})
                    if i == 0:
                        out = r
                self._sync_replicated()
                return out
            return on_every_shard
        return v

    def __setattr__(self, name, value):
        # (the re-armed blocks of the steady-state command are out of date)
        object.__setattr__(self, "_dirty", True)
        for s in object.__getattribute__(self, "_shards"):
            setattr(s, name, self._to(value, s))

    @staticmethod
    def _to(v, shard):
        if torch.is_tensor(v) and v.is_cuda and v.device != shard.d:
            return v.to(shard.d)
        return v

    def _sync_replicated(self):
        """the sequences every shard must hold identically are shard 0's (a constructor / reset() draw happens per
        device)"""
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

    # ---- one command
    # ---------------------------------------------------------------------------------------------------
    def command(self, state, shift_nominal_trajectory=True, info=None):
        """mppi.py:240-252 on N devices: the action (device 0), without synchronising."""
        shards = object.__getattribute__(self, "_shards")
        s0 = shards[0]
        shift = bool(shift_nominal_trajectory)
        if object.__getattribute__(self, "_fast") is not None:
            action = self._command_rearmed(state, shift, info)
            if action is not None:
                return action
        object.__getattribute__(self, "_replicas").sync()
        eng = object.__getattribute__(self, "_engine")
        # the state: ONCE to device 0 (a host state travels in a launch packet, MPPI._to_state); the other devices get
        # it by a peer copy in front of their K1, issued by their worker thread -- unless the shards take different rows
        # of it (per-sample initial states of the global problem, mppi.py:302-305) or there are no workers
        states, bc = [state] * len(shards), None
        if eng is not None:
            with _on(s0.d):
                st0 = s0._to_state(state) if tuple(getattr(state, "shape", ())) != (s0.K,
                        s0.nx) or s0.K_local == s0.K else None
            if st0 is not None:
                st0 = st0.contiguous()
                states, bc = self._state_copies(st0), st0
        ps = []
        try:
            for g, s in enumerate(shards):
                s.info = info
                with (_on(s.d) if not self._light(s) else contextlib.nullcontext()):
                    if s._jit_pending is not None:
                        s._adopt_background_model()
                    if getattr(s._model, "watch", None) is not None:
                        s._check_traced(state if s is s0 else None)
                    # (MPPI._to_state moves the state to the shard's device and, for per-sample initial states of the
                    # global problem -- (K, nx), mppi.py:302-305 --, takes this shard's rows)
                    ps.append(s._prepare(states[g], shift))
        except RuntimeError as e:
            if "Expected all tensors to be on the same device" in str(e):
                raise RuntimeError(f"{e}\n(pytorch_mppi_amd device group: every shard calls dynamics / running_cost "
                        f"with tensors on ITS "
                                   "device; torch.nn.Modules are copied there, anything else the callables read must "
                                           "follow "
                                   "`state.device` -- pytorch_mppi_amd/group.py)") from e
            raise
        # the blocks as prepared
        fresh = [N.MppiProblem.from_buffer_copy(p) for p in ps] if eng is not None else None
        handed_over = self._issue(ps, bc, states)
        self._arm(ps, fresh, states, bc, handed_over)
        action = None
        for s, p in zip(shards, ps):
            if type(s)._end is _plain_end:
                # (no launches in there: no need for the shard's device to be current)
                a = s._end(p)
            else:
                with _on(s.d):
                    a = s._end(p)
            if s is s0:
                action = a
        return action

    # ---- the steady-state command: the blocks of the previous command, re-armed
    # ---------------------------------------------
    # A control loop commands the same problem again and again: what changes between two commands of a plain MPPI group
    # on the engine's generator is the command counter, the nominal sequence (the previous command's result), the state
    # and where the results go.  `_prepare` rebuilds everything else too -- per shard a parameter key, a struct copy,
    # five allocations: ~25 us of Python, N times -- which at N = 8 is more than the 78 us a C3-sized command takes on
    # the GPUs.  So after a command that went through the workers, the group keeps the blocks AS PREPARED (`fresh`) and
    # two sets of result buffers per shard, and the
    # next command only writes those few fields into a copy of them and hands it over -- as long as nothing was assigned
    # or called on the group since (`_dirty`: every attribute write and method call sets it), shard 0's parameter key is
    # what it was (in-place edits of its parameter tensors move it) and the state is a single (nx,) vector.  Anything
    # else -- and every other controller class, noise mode or path -- takes the ordinary way above.  Shard 0's U stays a
    # NEW tensor per command (the returned action is a view of it and must never change later, mppi.py:270-275); the
    # other shards' sequences and the per-sample results live in the two buffer sets by command parity (the group's own
    # `cost_total` / `omega` reads concatenate
    # copies; the exchange's event chain keeps a set untouched until every device has finished reading it:
    # csrc/group.hip).
    def _arm(self, ps, fresh, states, bc, forms):
        object.__setattr__(self, "_fast", None)
        shards = object.__getattribute__(self, "_shards")
        if forms is None or fresh is None or bc is None or not object.__getattribute__(self, "_rearm"):
            return
        s0 = shards[0]
        if not all(type(s)._prepare is _plain_prepare and type(s)._end is _plain_end and self._light(s)
                and s.u_per_command >= 1
                   for s in shards) or tuple(bc.shape) != (s0.nx,):
            return
        bufs = []
        for g, s in enumerate(shards):
            mk = lambda *shape: [torch.empty(*shape, device=s.d, dtype=s.dtype) for _ in range(2)]
            bufs.append(dict(cost=mk(s.K_local), omega=mk(s.K_local), wnz=mk(s.K_local), record=mk(2 + s.T * s.nu),
                             U=mk(s.T, s.nu) if g > 0 else None))
        object.__setattr__(self, "_fast", dict(key0=s0._static_key(s0.T), fresh=fresh, live=ps, bufs=bufs,
                forms=list(forms), n=0,
                                               state_ptrs=[t.data_ptr() for t in states], state_shape=tuple(bc.shape),
                                               records=[p._keep["records"] for p in ps], size=C.sizeof(N.MppiProblem)))
        object.__setattr__(self, "_dirty", False)

    def _command_rearmed(self, state, shift, info):
        """one steady-state command (see above), or None: take the ordinary way"""
        f = object.__getattribute__(self, "_fast")
        shards = object.__getattribute__(self, "_shards")
        s0 = shards[0]
        if object.__getattribute__(self, "_dirty") or s0._static_key(s0.T) != f["key0"] or s0._injected:
            object.__setattr__(self, "_fast", None)
            return None
        if torch.is_tensor(state) and state.device == s0.d:
            st0 = s0._to_state(state)
        else:
            # (a host state travels in a launch packet on the CURRENT device)
            with _on(s0.d):
                st0 = s0._to_state(state)
        if tuple(st0.shape) != f["state_shape"] or not st0.is_contiguous():
            object.__setattr__(self, "_fast", None)
            return None
        eng = object.__getattribute__(self, "_engine")
        lib = N.lib()
        G = len(shards)
        par = f["n"] & 1
        f["n"] += 1
        streams = [torch._C._cuda_getCurrentRawStream(s._dev_index) for s in shards]
        ptr0 = st0.data_ptr()
        sp = f["state_ptrs"]
        if any(sp[g] != sp[0] for g in range(1, G)):
            dst = (C.c_void_p * G)(*[None if sp[g] == sp[0] else sp[g] for g in range(G)])
            N.check(lib.mppi_group_broadcast(eng, ptr0, st0.numel() * st0.element_size(), dst, streams[0]),
                    "mppi_group_broadcast")
        U0_new = torch.empty(s0.T, s0.nu, device=s0.d, dtype=s0.dtype)       # the action is a view of it: never reused
        outs = []
        for g, s in enumerate(shards):
            p, b = f["live"][g], f["bufs"][g]
            C.memmove(C.byref(p), C.byref(f["fresh"][g]), f["size"])           # the block as `_prepare` left it ...
            s._call += 1                                                        # ... with this command's few fields
            U_out = U0_new if g == 0 else b["U"][par]
            cost, omega, wnz, record = b["cost"][par], b["omega"][par], b["wnz"][par], b["record"][par]
            p.call, p.shift = s._call, int(shift)
            p.U, p.U_out = s.U.data_ptr(), U_out.data_ptr()
            p.state = ptr0 if sp[g] == sp[0] else sp[g]
            p.cost_total, p.omega = cost.data_ptr(), omega.data_ptr()
            p.cost_total_non_zero, p.record = wnz.data_ptr(), record.data_ptr()
            rc = lib.mppi_group_submit(eng, g, C.byref(p), None, f["records"][g].data_ptr(), streams[g])
            if rc != 0:
                lib.mppi_group_abort(eng)
                object.__setattr__(self, "_fast", None)
                N.check(rc, "mppi_group_submit")
            outs.append((U_out, cost, omega, wnz, record))
        forms = (C.c_int32 * G)()
        t_w = time.perf_counter()
        rc = lib.mppi_group_wait(eng, forms, None)
        object.__setattr__(self, "wait_seconds", object.__getattribute__(self, "wait_seconds") + time.perf_counter() - t_w)
        if rc != 0:
            object.__setattr__(self, "_fast", None)
            N.check(rc, "mppi_group_wait")
        if [int(x) for x in forms] != f["forms"]:
            # the engine took another form than last time: look again next command
            object.__setattr__(self, "_dirty", True)
        for g, s in enumerate(shards):
            p = f["live"][g]
            U_out, cost, omega, wnz, record = outs[g]
            k = p._keep
            k["U"], k["U_new"], k["omega"], k["wnz"], k["record"] = s.U, U_out, omega, wnz, record
            if g == 0:
                k["state"] = st0
            if p.noise_src == N.NOISE_PHILOX and p.z:
                p.noise_src = N.NOISE_TNK4                     # the rows K1 generated are in p.z now (lazy attributes)
            s.info, s.state, s.cost_total = info, st0 if g == 0 else s.state, cost
            s._states = s._actions = s._noise = s._perturbed_action = None
            s._omega, s._wnz, s._record, s._lazy_w, s._last, s.U = omega, wnz, record, None, p, U_out
        return U0_new[0] if s0.u_per_command == 1 else U0_new[:s0.u_per_command]

    @staticmethod
    def _light(s):
        """a shard whose `_prepare` launches nothing from this thread (the engine's generator inside K1, nothing to
        convert or
        upload): its device need not be made current for it"""
        return (s.rng == "philox" and s.last_draw in ("philox-onchip", "philox-k1") and not s._injected and s.M == 1
                and s.specific_action_sampler is None and type(s)._prepare is _plain_prepare and s._model is not None
                and getattr(s._model, "watch", None) is None and s._jit_pending is None)

    def _state_copies(self, st0):
        """per shard the tensor its K1 reads the state from: device 0's itself where the shard lives there, else a
        buffer on
        the shard's device that its worker fills from device 0's in front of K1 (mppi_group_broadcast)"""
        shards = object.__getattribute__(self, "_shards")
        bufs = object.__getattribute__(self, "_state_bufs")
        key = (tuple(st0.shape), st0.dtype)
        if bufs is None or bufs[0] != key:
            bufs = (key, [None if s.d == st0.device else torch.empty(st0.shape, dtype=st0.dtype,
                    device=s.d) for s in shards])
            object.__setattr__(self, "_state_bufs", bufs)
        return [st0 if b is None else b for b in bufs[1]]

    def _records_for(self, g, s, n):
        """the (G, 2 + J) buffer the exchange fills on device g: one per shard and record size, touched on that shard's
        stream only"""
        rb = object.__getattribute__(self, "_rec_bufs")
        b = rb.get((g, n, s.dtype))
        if b is None:
            if len(rb) > 4 * len(object.__getattribute__(self, "_shards")):
                rb.clear()
            b = rb[(g, n, s.dtype)] = torch.empty(len(object.__getattribute__(self, "_shards")), n, device=s.d,
                    dtype=s.dtype)
        return b

    def _issue(self, ps, bc, states):
        """the launches of a prepared command on every device + the exchange + K5"""
        shards = object.__getattribute__(self, "_shards")
        eng = object.__getattribute__(self, "_engine")
        lib = N.lib()
        G = len(shards)
        blocks = [s._group_blocks(p) if p._deferred else None for s, p in zip(shards, ps)]
        if eng is not None and all(b is not None for b in blocks):
            streams = [torch._C._cuda_getCurrentRawStream(s._dev_index) for s in shards]
            if bc is not None and any(t is not bc for t in states):
                dst = (C.c_void_p * G)(*[None if t is bc else t.data_ptr() for t in states])
                N.check(lib.mppi_group_broadcast(eng, bc.data_ptr(), bc.numel() * bc.element_size(), dst, streams[0]),
                        "mppi_group_broadcast")
            for g, (s, p, (b, bt)) in enumerate(zip(shards, ps, blocks)):
                q = bt if bt is not None else b
                rec = self._records_for(g, s, 2 + q.T * q.nu)
                p._keep["records"] = rec
                rc = lib.mppi_group_submit(eng, g, C.byref(b), C.byref(bt) if bt is not None else None, rec.data_ptr(),
                        streams[g])
                if rc != 0:
                    lib.mppi_group_abort(eng)
                    N.check(rc, "mppi_group_submit")
            forms, nds = (C.c_int32 * G)(), (C.c_int32 * G)()
            t_w = time.perf_counter()
            rc = lib.mppi_group_wait(eng, forms, nds)
            object.__setattr__(self, "wait_seconds", object.__getattribute__(self, "wait_seconds") + time.perf_counter() - t_w)
            if rc == 0:
                for g, (s, p) in enumerate(zip(shards, ps)):
                    s._launched(p, int(forms[g]), int(nds[g]))
                return [int(f_) for f_ in forms]
            if rc != N.E_UNSUPPORTED:
                N.check(rc, "mppi_group_wait")
            # no one-call form for this command on this model (the in-place (K,T,nu) K1, KMPPI's fused interpolation):
            # the shards
            # issue it themselves, their own way.  Nothing was exchanged or applied; what was launched is launched again
        for s, p in zip(shards, ps):
            if p._deferred:
                with _on(s.d):
                    s._launch_prepared(p)
        self._exchange(ps)
        return None

    def _exchange(self, ps):
        """records -> every device, K5 everywhere, from THIS thread (the form without worker threads)"""
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
                (C.c_void_p * G)(*[b.data_ptr() for b in bufs]), (C.c_void_p * G)(*streams)),
                        "mppi_exchange_combine_all")
            return
        # staged: the records travel by device copies (torch orders them against the streams involved), K5 on every
        # device
        for s, p in zip(shards, ps):
            with _on(s.d):
                s._combine(p, torch.stack([r if r.device == s.d else r.to(s.d) for r in recs]))


from .mppi import MPPI as _MPPI     # noqa: E402  (mppi.py imports this module lazily, inside MPPI.__new__)

_plain_end = _MPPI._end
_plain_prepare = _MPPI._prepare
