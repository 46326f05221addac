"""`MPPI` / `KMPPI` with the reference's constructor and `.command(state)` surface
(/root/reference/src/pytorch_mppi/mppi.py:35-448, :593-688), backed by the HIP engine.

Host code here is parameter resolution and launch plumbing only; the arithmetic of
`_compute_total_cost_batch` -> `_compute_weighting` -> weighted update runs in
csrc/*.hip through the C-ABI (include/mppi_amd.h).  Two ways into the engine:

* fused path  -- `dynamics`/`running_cost` are the bound methods of a `models.NativeModel`:
  one K1 launch does noise colouring, bounding, action cost, the T-step rollout and the
  running cost; K3/K4 do the exp-weighted update.  Nothing of shape (K,T,nu) is materialised
  unless a caller reads `noise` / `perturbed_action` / `states` (lazy).
* generic path -- any other callable (the reference's plugin API): `mppi_prepare` materialises
  the bounded actions, the T-loop calls the user's torch callables on device tensors exactly
  like mppi.py:312-322, then the same K3/K4.

Additive, keyword-only extras (not in the reference): ``rng`` ("torch": draw
``torch.randn(K,T,nu)`` like mppi.py:203 -- identical generator consumption and, on the same
device and seed, identical draws -- computed by the engine's own launch straight into its rows,
command n+1's draw inside command n's K3 launch where nobody else touches the generator in
between; "torch-native": the same generator drawn directly in the engine's sample-minor layout;
"philox": generate in-kernel, no (K,T,nu) array at all; "philox7": the same with Philox4x32-7),
``seed``, ``shard`` (multi-GPU, one process per GPU: dist.py), ``devices`` (multi-GPU from ONE
process: group.py), ``auto_jit`` (plain torch callables traced into device functors: trace.py).
"""
import ctypes as C
import logging
import os
import typing

import torch

from . import _native as N
from .models import MLPResidual, NativeModel, native_model_of

logger = logging.getLogger(__name__)

_DT = {torch.float32: N.F32, torch.float64: N.F64}
_TORCH_ROWS = {}     # rng="torch": (device, K, T, nu) -> did csrc/noise_torch.hip reproduce torch.randn bit for bit (MPPI._torch_stream_fill)


class SpecificActionSampler:
    """Same hook as mppi.py:16-32."""

    def __init__(self):
        self.start_idx = 0
        self.end_idx = 0
        self.slice = slice(0, 0)

    def sample_trajectories(self, state, info):
        raise NotImplementedError

    def specific_dynamics(self, next_state, state, action, t):
        return next_state

    def register_sample_start_end(self, start_idx, end_idx):
        self.start_idx = start_idx
        self.end_idx = end_idx
        self.slice = slice(start_idx, end_idx)


def _auto_jit_mode(v):
    """auto_jit / MPPI_AUTO_JIT -> "sync" | "async" | "0" (anything else is an error: `MPPI_AUTO_JIT=false` must not mean on)"""
    if isinstance(v, str):
        v = v.strip().lower()
    m = {True: "sync", "1": "sync", "sync": "sync", "true": "sync", "on": "sync", "yes": "sync", "async": "async",
         False: "0", "": "0", "0": "0", "off": "0", "false": "0", "no": "0", "none": "0"}.get(v)
    if m is None:
        raise ValueError(f"auto_jit / MPPI_AUTO_JIT = {v!r}: expected 'sync', 'async' or '0' (aliases: True/1/on, False/0/off/false/no)")
    return m


_TRACE = []


def _trace_module():
    """pytorch_mppi_amd.trace, imported on first use (an `import` statement per command costs a microsecond of a 15 us budget)"""
    if not _TRACE:
        from . import trace
        _TRACE.append(trace)
    return _TRACE[0]


def _ptr(t):
    # a plain int is what a ctypes c_void_p field wants; no wrapper object per pointer per command
    return None if t is None else t.data_ptr()


class MPPI:
    """Model Predictive Path Integral control (Williams et al. 2017, alg. 2), drop-in for
    `pytorch_mppi.MPPI` on MI355X."""

    def __new__(cls, *args, devices=None, **kw):
        # devices=[d0, d1, ...] (two or more): ONE Python process commanding on several GPUs -- the object is a device group
        # (pytorch_mppi_amd/group.py: one shard controller per device, a subclass of `cls`); SURVEY.md 8b / 8e
        if devices is not None and len(devices) > 1:
            from .group import DeviceGroup, group_class
            if not issubclass(cls, DeviceGroup):
                return object.__new__(group_class(cls))
        return object.__new__(cls)

    def __init__(self, dynamics, running_cost, nx, noise_sigma, num_samples=100, horizon=15, device="cpu",
                 terminal_state_cost=None,
                 lambda_=1.,
                 noise_mu=None,
                 u_min=None,
                 u_max=None,
                 u_init=None,
                 U_init=None,
                 u_scale=1,
                 u_per_command=1,
                 step_dependent_dynamics=False,
                 rollout_samples=1,
                 rollout_var_cost=0,
                 rollout_var_discount=0.95,
                 sample_null_action=False,
                 specific_action_sampler: typing.Optional[SpecificActionSampler] = None,
                 noise_abs_cost=False,
                 *, rng="torch", seed=None, shard=None, auto_jit=None, devices=None):
        if devices is not None:
            if len(devices) != 1:
                raise ValueError("devices= needs at least one device")       # (two or more never get here: __new__)
            device = torch.device("cuda", devices[0]) if isinstance(devices[0], int) else torch.device(devices[0])
        self.d = torch.device(device) if not isinstance(device, torch.device) else device
        self.dtype = noise_sigma.dtype                                   # mppi.py:88
        if self.dtype not in _DT:
            raise TypeError(f"noise_sigma dtype {self.dtype} unsupported (float32/float64)")
        self.K = num_samples
        self.T = horizon
        self.nx = nx
        self.nu = 1 if len(noise_sigma.shape) == 0 else noise_sigma.shape[0]   # :94
        self.lambda_ = lambda_

        if noise_mu is None:
            noise_mu = torch.zeros(self.nu, dtype=self.dtype)
        if u_init is None:
            u_init = torch.zeros_like(noise_mu)
        if self.nu == 1:                                                  # :104-106
            noise_mu = noise_mu.view(-1)
            noise_sigma = noise_sigma.view(-1, 1)

        self.u_scale = u_scale
        self.u_per_command = u_per_command
        if u_max is not None and u_min is None:                           # :112-119
            if not torch.is_tensor(u_max):
                u_max = torch.tensor(u_max)
            u_min = -u_max
        if u_min is not None and u_max is None:
            if not torch.is_tensor(u_min):
                u_min = torch.tensor(u_min)
            u_max = -u_min
        if u_min is not None:                                             # :121-126
            self.u_min = u_min.to(device=self.d)
            self.u_max = u_max.to(device=self.d)
        else:
            self.u_min = torch.tensor(float('-inf'), device=self.d)
            self.u_max = torch.tensor(float('inf'), device=self.d)

        self.noise_mu = noise_mu.to(self.d)
        self.noise_sigma = noise_sigma.to(self.d)
        self._refresh_noise_factors()                                     # :130-139
        # shard = (rank, world_size[, process_group]): this controller holds samples
        # [k_offset, k_offset + K_local) of the K global ones (dist.py)
        self._shard = None
        self._shard_gen = None
        self.k_offset = 0
        self.K_local = self.K
        if shard is not None:
            from .dist import ShardPlan
            self._shard = ShardPlan(self.K, *shard)
            self.k_offset = self._shard.k_offset
            self.K_local = self._shard.K_local
        self.U = U_init
        self.u_init = u_init.to(self.d)
        if self.U is None:
            self.U = self._replicated(self._sample_noise((self.T,)))      # :144-145
        else:
            self.U = self.U.to(device=self.d, dtype=self.dtype)

        self.step_dependency = step_dependent_dynamics
        if step_dependent_dynamics:                                       # :147-154
            self._dynamics_fn = dynamics
            self._running_cost_fn = running_cost
        else:
            self._dynamics_fn = lambda state, u, t: dynamics(state, u)
            self._running_cost_fn = lambda state, u, t: running_cost(state, u)
        self.F = dynamics
        self.running_cost = running_cost
        self.terminal_state_cost = terminal_state_cost
        self.sample_null_action = sample_null_action
        self.specific_action_sampler = specific_action_sampler
        self._terminal_state_cost_fn = terminal_state_cost if terminal_state_cost is not None \
            else (lambda states, actions: 0)
        self.noise_abs_cost = noise_abs_cost
        self.state = None
        self.info = None

        self.M = rollout_samples
        self.rollout_var_cost = rollout_var_cost
        self.rollout_var_discount = rollout_var_discount
        if self.M > 1:
            self._var_discount_factors = rollout_var_discount ** torch.arange(
                self.T, device=self.d, dtype=self.dtype)
        else:
            self._var_discount_factors = None

        # results of the last command (mppi.py:180-184)
        self.cost_total = None
        self._omega = None
        self._wnz = None
        self._lazy_w = None        # (lambda used, record) when omega / cost_total_non_zero are derived on first read
        self._states = None
        self._actions = None
        self._noise = None
        self._perturbed_action = None
        self._last = None          # what the lazy attributes need to re-derive (K,T,nu) arrays

        # ---- engine state ----
        if rng not in ("torch", "torch-native", "philox", "philox7"):
            raise ValueError("rng must be 'torch', 'torch-native', 'philox' or 'philox7'")
        # rng="philox7": the engine's generator with Philox4x32-7 (Random123's philox4x32_R<7>: the fewest rounds that pass BigCrush)
        # instead of -10 -- another stream, everything else as rng="philox"; 30 % fewer of the multiplies the on-chip command's
        # time is made of (MppiProblem.philox_rounds; oracle/philox.py `rounds`)
        self.philox_rounds = 7 if rng == "philox7" else 10
        self.rng = rng = "philox" if rng == "philox7" else rng
        self.philox_store = True   # rng="philox": K1 stores the generated rows, K3 re-reads them
        # sharded + Philox generator launch: queue the next command's rows behind K4 so that they
        # run while the record all-gather is in flight.  OFF: on this stack a kernel on torch's
        # default stream and one on a pool stream (RCCL's) do not run concurrently (measured: 42.8 us
        # spin kernel on a side stream + 32.3 us generator on the default stream = 72.2 us), so there is
        # nothing to win and the fork/join costs 13 us per command.
        self.overlap_collective = False
        self._pf_rows = None       # sharded + Philox: (key, rows) generated ahead for the next command
        self._pf_hits = 0
        self.last_draw = None      # how the last command got its normals: "philox-fill" | "philox-k1" | None (other modes)
        self.coloured_fill = True  # rng="philox", full Sigma: let the generator launch apply chol(Sigma) (see _draw_noise)
        self.philox_fill = None    # rng="philox": generate in a separate launch (True) / inside K1 (False) / by horizon (None)
        # rng="philox": the on-chip command (csrc/rollout_onchip.hpp) -- no (K,T,nu) array at all: one launch generates,
        # rolls out, keeps the bounded noise in accumulation registers / LDS and leaves one partial record per workgroup,
        # a second one combines them.  None: whenever the problem is in its scope (fp32, diagonal Sigma, plain MPPI,
        # M = 1, no sampler rows) and too large for the single-launch form; True / False: force / forbid.
        # (a full Sigma is coloured in the lane: L z + mu per timestep out of LDS)
        self.philox_onchip = None
        self._onchip_refused = False
        self.ktn_direct = os.environ.get("MPPI_KTN_DIRECT", "1") != "0"   # rng="torch": read (K,T,nu) in place when possible
        # the on-chip command: let what fits neither registers nor LDS wait in memory (one array per controller, allocated on first
        # use) instead of generating it twice
        self.onchip_spill = os.environ.get("MPPI_ONCHIP_SPILL", "1") != "0"
        self._spill = None
        # rng="torch": compute torch.randn's values straight into the engine's rows (see _torch_stream_fill); off: call
        # torch.randn and read / convert its (K,T,nu) array
        self.torch_rows = os.environ.get("MPPI_TORCH_ROWS", "1") != "0"
        # ... and the NEXT command's draw inside this command's K3 launch (ABI 21; adopted at the next command when the generator
        # is where that assumed: _torch_stream_fill).  Costs a second row buffer
        self.draw_ahead = os.environ.get("MPPI_DRAW_AHEAD", "1") != "0"
        # (draws of fewer normals than this keep their own tiny launch: carving them into K3's few workgroups costs more than it saves
        # -- profiles/r05_small_k_sweep.txt)
        self.draw_ahead_min = int(os.environ.get("MPPI_DRAW_AHEAD_MIN", str(1 << 19)))
        # the same for the ENGINE's generator (rng="philox" with rows in memory, MPPI_NEXT_DRAW_PHILOX): built, bit-exact, and OFF --
        # that generator launch is already bound by its 201 MB of stores (34 us at C3), not by the VALU, and a launch that reads K3's
        # rows while it writes the next ones moves the same 403 MB slower (mixed traffic: 5.3 TB/s against 5.9 one after the other;
        # C3 rows-in-memory command 0.1186 ms with, 0.1096 without; profiles/r05_draw_ahead_forms.txt)
        self.draw_ahead_philox = os.environ.get("MPPI_DRAW_AHEAD_PHILOX", "0") == "1"
        self._next_draw = None         # (shape key, generator, seed, offset, rows): generated, waiting for the next command
        self._next_armed = None        # ... handed to the engine with this command, not yet confirmed (_settle_next)
        self._next_hits = self._next_misses = self._next_cmds = 0
        self._zbuf_alt = {}
        self._generic_memo = None
        self._in_capture = False
        self._force_collective = False
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self._call = 0
        self._injected = []
        self._model = None
        m = native_model_of(dynamics, running_cost, terminal_state_cost)
        # step-dependent callbacks (mppi.py:147-154): fused when the native model's callables take t too
        # (jit.compile_model(..., step_dependent=True); the device functor always sees the timestep)
        # the built-in models ignore t (step_dependent None: either setting)
        if m is not None and (getattr(m, "step_dependent", False) is None
                              or bool(step_dependent_dynamics) == bool(getattr(m, "step_dependent", False))):
            self._model = m
        # a model that jit.from_torch traced from plain callables: re-traces (watch, spot-check) run on those callables
        self._traced_user_callables = ((m._dyn, m._cost, m._term) if self._model is not None and getattr(m, "_code", None) is not None
                                       and getattr(m, "watch", None) is not None else None)
        self.jit_note = None
        self._jit_pending = None
        # auto_jit: True / "sync" = trace and compile now (construction blocks for the hipcc run unless the object is cached);
        # "async" = trace now, compile in a background thread -- commands run the callbacks until the fused kernels are
        # there; False / "0" = off.  None: the environment's MPPI_AUTO_JIT (default "async")
        mode = _auto_jit_mode(auto_jit if auto_jit is not None else os.environ.get("MPPI_AUTO_JIT", "async"))
        # traced callables are re-checked against the live ones: a flat watch of the places they can read from on every
        # command (watch.StateWatch), and functor-against-callables on a small random batch on the device at adoption and
        # every MPPI_JIT_CHECK_EVERY commands (default 256; 0 = never) -- _check_traced / _spot_check below
        self._jit_mode = mode
        self._jit_check_every = int(os.environ.get("MPPI_JIT_CHECK_EVERY", "256"))
        self._jit_cmds = 0             # fused commands since the current traced model was adopted
        # share of the issuing time the spot-checks may take: the interval is stretched beyond `_jit_check_every` where a check
        # (~1 ms) would cost more than this (0: never stretched)
        self._jit_check_share = float(os.environ.get("MPPI_JIT_CHECK_SHARE", "0.01"))
        self._jit_next_check = 0       # ... and the command at which the next on-device spot-check is due
        self._jit_last_check = None    # (command number, time) of the previous one
        self._jit_retraces = 0         # times the callables' state moved in a way that changed the functor
        self._jit_benign = 0           # ... in a way that did not
        self._jit_spot_checks = 0
        self._jit_dynamic = []         # places (watch.Path) whose tensors were seen to change: run-time parameters from then on
        if m is None and self.d.type == "cuda" and self.M == 1 and mode != "0":
            # plain torch callables (the reference's plugin API): try to trace them into a device functor
            # (pytorch_mppi_amd/trace.py -> jit.compile_model); outside the traceable subset the generic path stays
            self._model = self._try_trace(dynamics, running_cost, terminal_state_cost, bool(step_dependent_dynamics),
                                          background=(mode == "async"))
        if self._model is not None and (self._model.nx != self.nx or self._model.nu != self.nu):
            raise ValueError(f"native model dims ({self._model.nx},{self._model.nu}) != (nx,nu)=({self.nx},{self.nu})")
        if self._shard is not None and self._shard.world_size > 1 and rng != "philox":
            # torch-generator modes: a shared U needs identically seeded ranks, which would make every
            # shard draw the SAME perturbations (effective samples K / world).  Each shard therefore
            # draws its rows from its own generator, keyed by (seed, rank); the default generator
            # stays in lock-step across ranks (it only feeds the replicated U draws).
            self._shard_gen = torch.Generator(device=self.d)
            self._shard_gen.manual_seed((self.seed + 0x9E3779B97F4A7C15 * (self._shard.rank + 1)) & 0x7FFFFFFFFFFFFFFF)
        self._ws = None
        self._zbuf = {}
        self._rec_buf = None
        self._vec_cache = {}
        self._problem_cache = {}
        self._ws_need = {}
        self._dev_index = (self.d.index if self.d.index is not None else
                           (torch.cuda.current_device() if self.d.type == "cuda" and torch.cuda.is_available() else 0))

    def _try_trace(self, dynamics, running_cost, terminal_state_cost, step_dependent, background=False, dynamic=(),
                   verify_in_background=False):
        """Plain torch callables -> fused model, or None (generic path; `jit_note` says why).  background: the hipcc run
        happens in a thread unless the object is cached; verify_in_background: so does the host check of the trace (a
        RE-trace in the middle of a control loop must not stall it for the second g++ takes)."""
        import logging
        from . import jit, trace, watch
        log = logging.getLogger("pytorch_mppi_amd")
        try:
            # the places the callables can read from, snapshotted BEFORE they run on symbols: whatever they write there
            # themselves (call counters, `self.last = state`) shows up as a difference and is dropped in _settle_watch
            w = watch.StateWatch([dynamics, running_cost, terminal_state_cost])
            code = jit.trace_and_verify(dynamics, running_cost, self.nx, self.nu, terminal_state_cost, step_dependent, horizon=self.T,
                                        device=self.d, dtype=self.dtype, dynamic=dynamic, verify=not verify_in_background)
            w.forget([src.path for src, _ in code["param_tensors"] if isinstance(src, trace.PathParam)])
            cached = jit.traced_is_cached(code, self.nx, self.nu, dtype=self.dtype)
            if verify_in_background or (background and not cached):
                # the hipcc run (30 s - 2 min) happens beside the control loop: callbacks until it has finished
                import threading
                box = {"watch": w}

                def work():
                    try:
                        if verify_in_background:
                            jit.verify_traced(code, dynamics, running_cost, self.nx, self.nu, terminal_state_cost, step_dependent, self.T)
                        box["model"] = jit.compile_traced(code, dynamics, running_cost, self.nx, self.nu, terminal_state_cost,
                                                          step_dependent=step_dependent, dtype=self.dtype)
                    except Exception as e:                      # a failed check / hipcc run: stay on the callbacks
                        box["error"] = e
                th = threading.Thread(target=work, name="pytorch_mppi_amd-jit", daemon=True)
                self._jit_pending = (th, box)
                th.start()
                if not (verify_in_background and cached):
                    self.jit_note = "generic path for now: the fused kernels of the traced callables are being compiled in the background"
                    log.warning("pytorch_mppi_amd: %s (auto_jit='sync' / MPPI_AUTO_JIT=sync waits for them instead)", self.jit_note)
                return None
            m = jit.compile_traced(code, dynamics, running_cost, self.nx, self.nu, terminal_state_cost, step_dependent=step_dependent,
                                   dtype=self.dtype)
        except trace.TraceUnsupported as e:
            self.jit_note = f"generic path: {e}"
            log.info("pytorch_mppi_amd: dynamics / running_cost stay on the generic (callback) path: %s", e)
            return None
        except Exception as e:           # a callable that fails on symbolic inputs in its own way, a failed hipcc run, ...
            self.jit_note = f"generic path: {type(e).__name__}: {e}"
            log.info("pytorch_mppi_amd: dynamics / running_cost stay on the generic (callback) path: %s: %s", type(e).__name__, e)
            return None
        m.watch = w
        self._settle_watch(m)            # (what the callables wrote to their own state while they were traced and checked)
        self._jit_cmds = 0
        self._jit_next_check, self._jit_last_check = 0, None
        self.jit_note = f"fused: traced {m.traced_ops} operations per sample into {m.name}"
        log.info("pytorch_mppi_amd: %s", self.jit_note)
        return m

    # -- traced callables against the live ones ---------------------------------------------------------------------------
    def _callables(self):
        raw = self._traced_user_callables
        if raw is not None:                       # a jit.from_torch model: the user's own callables, not the model's wrappers
            return raw[0], raw[1], raw[2], bool(self.step_dependency)
        return self.F, self.running_cost, self.terminal_state_cost, bool(self.step_dependency)

    def _drop_traced(self, note):
        import logging
        self.jit_note = note
        logging.getLogger("pytorch_mppi_amd").warning("pytorch_mppi_amd: %s", note)
        self._model = None
        self._problem_cache.clear()

    def _check_traced(self, state=None):
        """Once per command, before anything is launched (mppi.py:314,318 call the user's callables on every command: a
        traced functor has to notice when they would now compute something else).  Cheap part, every command: the
        parameter tensors' version counters (`refresh_params`), the version counters of tensors that became constants,
        and the watch over every place the callables can read from (watch.StateWatch.changed, ~0.1 us per place).
        Every `_jit_check_every` commands and on the first command of a newly adopted model: `_spot_check`."""
        m = self._model
        w = getattr(m, "watch", None)
        if w is None:
            return
        trace = _trace_module()
        try:
            if m._param_tensors:
                m.refresh_params()
            moved = w.changed()
        except trace.StaleTrace as e:
            moved, w = None, None
            self._traced_state_moved([], str(e))
            return
        if moved or (m._captured and m.stale()):
            self._traced_state_moved(moved, None)
            return
        n = self._jit_cmds
        self._jit_cmds = n + 1
        if self._jit_check_every > 0 and n >= self._jit_next_check and state is not None \
                and not torch.cuda.is_current_stream_capturing():
            # every `_jit_check_every` commands -- stretched, for problems so small that a check (a millisecond: the user's
            # callables on a batch, a tiny fused rollout, one device sync) would cost more than `_jit_check_share` (1 %) of the
            # time between two of them, to that many commands: a 20 us command is checked every ~6000 commands = 0.12 s
            import time
            t0 = time.perf_counter()
            ok = self._spot_check(state)
            t1 = time.perf_counter()
            last = self._jit_last_check
            every = self._jit_check_every
            if last is not None and n > last[0]:
                period = (t0 - last[1]) / (n - last[0])                 # seconds per command since the previous check
                if period > 0 and self._jit_check_share > 0:
                    every = max(every, min(65536, int((t1 - t0) / (self._jit_check_share * period))))
            self._jit_last_check = (n, t1)
            self._jit_next_check = n + every
            if ok:
                self._jit_unexplained = 0
            else:
                self._traced_state_moved([], "the fused functor and the callables disagree on a random batch")

    def _traced_state_moved(self, moved, why):
        """Something the traced callables can read is not what it was.  Re-trace (symbolic: milliseconds) and compare:
        the same functor source and parameter sources -> irrelevant (forget the places); the same source, parameters read
        from other tensors (a sub-module replaced by one of the same architecture) -> re-bind, no compile; anything else
        -> the fused kernels are out of date: back to the callables NOW (the reference's behaviour), new functor compiled
        beside the loop with the tensors that moved as run-time parameters."""
        from . import jit, trace
        m, w = self._model, self._model.watch
        dyn, rc, term, sd = self._callables()
        what = why or ("changed: " + w.describe(moved) if moved else "a tensor the traced callables read was modified in place")
        try:
            code = jit.trace_and_verify(dyn, rc, self.nx, self.nu, term, sd, verify=False, horizon=self.T, device=self.d,
                                        dtype=self.dtype, dynamic=self._jit_dynamic)
        except Exception as e:
            self._drop_traced(f"generic path: the callables' state changed ({what}) and they can no longer be traced: {type(e).__name__}: {e}")
            return
        if trace.same_functor(code, m._code) and why is None:
            self._jit_benign += 1
            if not trace.same_param_sources(code, m._code):
                m.rebind_params(code)
                self._problem_cache.clear()
            self._settle_moved(m, moved, code)
            return
        if why is not None and trace.same_functor(code, m._code) and trace.same_param_sources(code, m._code):
            # a spot-check mismatch that a fresh trace does not explain (a discontinuous cost on a boundary sample, state
            # behind a C extension, a tracer bug): the parameters were re-gathered by the spot-check.  Said aloud, and after
            # three in a row the controller stops trusting the functor: back to the callables, the reference's behaviour
            # (this is the case the spot-check exists for)
            self._jit_benign += 1
            self._jit_unexplained = getattr(self, "_jit_unexplained", 0) + 1
            import logging
            logging.getLogger("pytorch_mppi_amd").warning(
                "pytorch_mppi_amd: %s -- and a fresh trace of the callables prints the same functor (%d in a row)", why, self._jit_unexplained)
            if self._jit_unexplained >= 3:
                self._drop_traced("generic path: the fused functor disagreed with the callables on three spot-checks in a row and a fresh "
                                  "trace does not explain it (state behind a C extension? a tracer bug?); ctrl.retrace() tries again")
            else:
                self.jit_note = f"fused, but a spot-check found an unexplained mismatch ({self._jit_unexplained}): {m.name}"
            return
        self._jit_retraces += 1
        for path in w.tensors_at(moved or []):
            if not any(q.holder is path.holder and q.key == path.key for q in self._jit_dynamic):
                self._jit_dynamic.append(path)
        self._drop_traced(f"generic path for now: the callables' state changed ({what}); tracing them again")
        if self._jit_retraces > 16:
            self.jit_note = (f"generic path: the callables' state changed {self._jit_retraces} times in ways that change the functor; "
                             f"ctrl.retrace() tries again")
            return
        self._model = self._try_trace(dyn, rc, term, sd, background=True, dynamic=self._jit_dynamic, verify_in_background=True)

    def retrace(self, wait=True):
        """Trace the callables again now (what the controller does by itself when it sees their state move); wait=True
        blocks for the host check and the hipcc run unless the kernels are cached.  True when the controller runs fused."""
        if self._jit_pending is not None:
            self._jit_pending[0].join()
            self._jit_pending = None
        dyn, rc, term, sd = self._callables()
        if native_model_of(dyn, rc, term) is not None:
            return self._model is not None
        self._model = None
        self._problem_cache.clear()
        self._jit_retraces = 0
        self._model = self._try_trace(dyn, rc, term, sd, background=not wait, dynamic=self._jit_dynamic)
        if self._model is not None:
            self._settle_watch(self._model)
        return self._model is not None

    def _settle_watch(self, m):
        """A traced model is about to serve commands: places that moved since its watch was taken are either the callables'
        own doing (they ran on symbols and on the verification batches since) or a real change during the compile."""
        w = m.watch
        moved = w.changed()
        if not moved:
            return True
        from . import jit, trace
        dyn, rc, term, sd = self._callables()
        try:
            code = jit.trace_and_verify(dyn, rc, self.nx, self.nu, term, sd, verify=False, horizon=self.T, device=self.d,
                                        dtype=self.dtype, dynamic=m._code.get("dynamic", ()))
        except Exception:
            return False
        if not trace.same_functor(code, m._code):
            return False
        if not trace.same_param_sources(code, m._code):
            m.rebind_params(code)
        self._settle_moved(m, moved, code, adoption=True)
        return True

    def _settle_moved(self, m, moved, code, adoption=False):
        """A fresh trace prints the same functor although these watched places moved.  Which of them may be forgotten?  Only
        those the trace did NOT read (`cost.goal = torch.tensor([2., 1.])` -- same values, a new object, what a
        planner does every cycle -- was judged benign and `GoalCost.goal` dropped from the watch for good; the next, real
        change of the goal then went unseen).  A place keeps being watched, with its present value as the new reference, when
        that value is a tensor / array among the roots of the trace's constants or its parameter tensors, or a number / string
        equal to one of the graph's numeric constants; integers, booleans and strings (what Python-level control flow reads
        without leaving a constant behind) are forgotten only after three benign moves in a row, or at adoption (what moved
        while the callables were being traced and verified is their own bookkeeping).  A place re-bound to a NEW container or
        object gets the watch rebuilt over the roots, so that what hangs below the new object is watched too."""
        import numpy as np
        from . import trace, watch as watch_mod
        w = m.watch
        read = [c for c, _ in code["captured"]]
        for src, _ in code["param_tensors"]:
            try:
                read.append(trace.param_tensor(src))
            except Exception:
                pass
        numbers = code.get("numbers", frozenset())
        drop, keep, rebuild = [], [], False
        for i in moved:
            path = w.places[i][0]
            v = path.get()
            key = (id(path.holder), path.key if not isinstance(path.holder, watch_mod._Len) else "#len")
            if isinstance(v, (torch.Tensor, np.ndarray)):
                was_read = any(v is r for r in read)
                (keep if was_read else drop).append(i)
                continue
            if isinstance(v, (bool, str)) or (isinstance(v, int) and not isinstance(path.holder, watch_mod._Len)):
                n = w.benign.get(key, 0) + 1
                w.benign[key] = n
                in_graph = isinstance(v, (int, bool)) and float(v) in numbers
                (drop if (adoption or n >= 3) and not in_graph else keep).append(i)
                continue
            if isinstance(v, float):
                (keep if v in numbers else drop).append(i)
                continue
            if v is watch_mod._MISSING or v is None or isinstance(path.holder, watch_mod._Len) or isinstance(v, watch_mod._PRIMS):
                drop.append(i)
                continue
            # re-bound to another container / object: the trace may have read what hangs below it
            keep.append(i)
            rebuild = True
        if keep:
            w.resnap(keep)          # (indices stay valid: resnap replaces in place)
        if drop:
            w.drop(drop)
        if rebuild:
            dyn, rc, term, _ = self._callables()
            nw = watch_mod.StateWatch([dyn, rc, term])
            nw.benign = w.benign
            nw.forget(w.dropped_paths)
            nw.dropped, nw.dropped_paths = w.dropped, list(w.dropped_paths)
            m.watch = nw
        m._code = code
        m._captured = list(code["captured"])

    def _spot_check(self, state, samples=64, steps=4):
        """The fused functor against the user's callables on a small random batch ON THE DEVICE (`samples` states around the
        current one, `steps` timesteps of random bounded actions): total costs and visited states of a tiny fused rollout
        against the reference's own loop (mppi.py:297-332) over the same actions.  What the watch cannot see ends here:
        writes through `.data`, state behind C extensions, a tracer bug the host check did not meet.  One device sync."""
        m = self._model
        self._jit_spot_checks += 1
        try:
            m.refresh_params(force=True)              # (a write through .data moves no version counter)
        except Exception:
            return False
        Tp = max(1, min(int(steps), self.T))
        pr = getattr(m, "_probe", None)
        if pr is None or pr.T != Tp:
            term = m.terminal_state_cost if self.terminal_state_cost is not None else None
            pr = MPPI(m.dynamics, m.running_cost, self.nx, self.noise_sigma.reshape(self.nu, self.nu), num_samples=samples, horizon=Tp,
                      device=self.d, terminal_state_cost=term, lambda_=1.0, u_min=self.u_min, u_max=self.u_max, u_scale=self.u_scale,
                      step_dependent_dynamics=bool(self.step_dependency), U_init=torch.zeros(Tp, self.nu, dtype=self.dtype),
                      rng="torch", auto_jit=False)
            pr._want_states = True
            pr._jit_check_every = 0                    # (the probe runs the model under test: it does not check itself)
            m._probe = pr
        if pr._needs_generic() or getattr(m, "_spot_unavailable", False):
            return True                                # no fused kernel for this model at the probe's shape: nothing to compare
        gen = getattr(self, "_spot_gen", None)
        if gen is None:
            gen = self._spot_gen = torch.Generator(device=self.d)
            gen.manual_seed(0x5EED)
        x = self._to_state(state).reshape(-1)
        x = x[:self.nx] if x.numel() >= self.nx else torch.zeros(self.nx, device=self.d, dtype=self.dtype)
        X0 = x + torch.randn(samples, self.nx, device=self.d, dtype=self.dtype, generator=gen) * (0.5 * x.abs() + 1.0)
        z = torch.randn(samples, Tp, self.nu, device=self.d, dtype=self.dtype, generator=gen)
        with torch.no_grad():
            pr.U = torch.zeros(Tp, self.nu, device=self.d, dtype=self.dtype)     # U = 0: no action cost, cost_total is the rollout's
            pr.inject_noise(z)
            pr.command(X0, shift_nominal_trajectory=False)
            fused_c, fused_x, pa = pr.cost_total, pr.states, pr.perturbed_action

            def reference(dev):
                """the reference's own loop (mppi.py:297-332) over the same actions, its tensors on `dev`"""
                state, ref_c = X0.to(dev).clone(), torch.zeros(samples, device=dev, dtype=self.dtype)
                states = torch.empty(1, samples, Tp, self.nx, device=dev, dtype=self.dtype)
                actions = torch.empty(1, samples, Tp, self.nu, device=dev, dtype=self.dtype)
                pad = pa.to(dev)
                for t in range(Tp):
                    u = self.u_scale * pad[:, t]
                    state = self._dynamics_fn(state, u, t)
                    ref_c = ref_c + self._running_cost_fn(state, u, t).reshape(samples)
                    states[0, :, t] = state[:, :self.nx]
                    actions[0, :, t] = u
                if self.terminal_state_cost is not None:
                    c = self._terminal_state_cost_fn(states, actions)
                    ref_c = ref_c + (c.squeeze(0) if torch.is_tensor(c) and c.dim() > 1 else c)
                return ref_c.to(self.d), states.to(self.d)
            ref = None
            for dev in (self.d, torch.device("cpu")):
                # callables that only work on host tensors (numpy ufuncs on tensors: the reference's own pendulum,
                # tests/pendulum.py:45-46) are checked there; ones that work on neither cannot be checked at all
                try:
                    ref = reference(dev)
                    break
                except Exception:
                    continue
            if ref is None:
                m._spot_unavailable = True
                return True
            ref_c, states = ref
            tol = 2e-3 if self.dtype == torch.float32 else 1e-7
            bad = torch.zeros(samples, dtype=torch.bool, device=self.d)
            for got, ref in ((fused_c, ref_c), (fused_x[0].reshape(samples, -1), states[0].reshape(samples, -1))):
                got, ref = got.reshape(samples, -1), ref.reshape(samples, -1).to(got.dtype)
                fin = torch.isfinite(ref)
                scale = torch.where(fin, ref.abs(), torch.zeros_like(ref)).amax().clamp_min(1.0)
                d = torch.where(fin, (got - ref).abs(), torch.zeros_like(ref))
                bad |= ((d > tol * scale) | (fin != torch.isfinite(got))).any(dim=1)
            # more than a few samples off: not a boundary case of a discontinuous cost
            return int(bad.sum().item()) <= samples // 16

    # ------------------------------------------------------------------------------------------
    # parameter resolution (host, once per change)
    # ------------------------------------------------------------------------------------------
    def _refresh_noise_factors(self):
        """mppi.py:130-139.  Also packs the (nu,nu) factor the kernels read: chol(Sigma), or
        diag(sqrt(diag Sigma)) when Sigma is diagonal."""
        self._diagonal_sigma = torch.equal(self.noise_sigma, torch.diag(torch.diag(self.noise_sigma)))
        if self._diagonal_sigma:
            diag = torch.diag(self.noise_sigma)
            self._noise_sigma_inv_diag = 1.0 / diag
            self._noise_sigma_sqrt_diag = torch.sqrt(diag)
            self.noise_sigma_inv = torch.diag(self._noise_sigma_inv_diag)
            self._noise_L = torch.diag(self._noise_sigma_sqrt_diag).contiguous()
        else:
            self.noise_sigma_inv = torch.linalg.inv(self.noise_sigma)
            self._noise_sigma_chol = torch.linalg.cholesky(self.noise_sigma)
            self._noise_L = self._noise_sigma_chol.contiguous()
        # what the kernels read.  Like the reference's action-cost closure (mppi.py:189-199, values
        # captured at construction), later assignments to the PUBLIC `noise_sigma` /
        # `noise_sigma_inv` attributes (reference autotune.py:158-162) do not reach the hot path;
        # `set_noise()` is the coherent way to change Sigma.
        self._sigma_inv_kernel = self.noise_sigma_inv

    def set_noise(self, noise_sigma=None, noise_mu=None):
        """Replace Sigma / mu and refresh every derived factor (in the reference
        autotune rewrites `noise_sigma` but the sampler keeps the init-time factors)."""
        if noise_sigma is not None:
            s = torch.as_tensor(noise_sigma, dtype=self.dtype).to(self.d)
            self.noise_sigma = s.view(-1, 1) if self.nu == 1 else s
            self._refresh_noise_factors()
        if noise_mu is not None:
            self.noise_mu = torch.as_tensor(noise_mu, dtype=self.dtype).to(self.d).view(-1)

    def _sample_noise(self, shape):
        """mppi.py:201-206 -- only used for the (T,nu) initial / reset sequence."""
        z = torch.randn(*shape, self.nu, device=self.d, dtype=self.dtype)
        if self._diagonal_sigma:
            return z * self._noise_sigma_sqrt_diag + self.noise_mu
        return z @ self._noise_sigma_chol.T + self.noise_mu

    def compile(self, **kwargs):
        """mppi.py:208-215.  The fused path is already compiled HIP; on the generic path the
        user's callbacks are handed to torch.compile exactly like the reference."""
        if self._model is None:
            self._dynamics_fn = torch.compile(self._dynamics_fn, **kwargs)
            self._running_cost_fn = torch.compile(self._running_cost_fn, **kwargs)

    def get_params(self):
        return f"K={self.K} T={self.T} M={self.M} lambda={self.lambda_} noise_mu={self.noise_mu.cpu().numpy()} noise_sigma={self.noise_sigma.cpu().numpy()}".replace(
            "\n", ",")

    def get_action_sequence(self):
        return self.U

    def shift_nominal_trajectory(self):
        """mppi.py:232-238 (explicit call; `command` folds the shift into the kernels' reads)."""
        self.U = torch.roll(self.U, -1, dims=0)
        self.U[-1] = self.u_init

    def change_horizon(self, horizon):
        if horizon < self.U.shape[0]:
            self.U = self.U[:horizon]
        elif horizon > self.U.shape[0]:
            self.U = torch.cat((self.U, self.u_init.repeat(horizon - self.U.shape[0], 1)))
        self.T = horizon
        self._ws = None
        self._problem_cache = {}

    def reset(self):
        self.U = self._replicated(self._sample_noise((self.T,)))

    def _replicated(self, t):
        """Sharded controllers: a tensor every rank must hold identically (the randomly initialised
        nominal sequence, mppi.py:144-145 / :290) is rank 0's draw, broadcast.  No process group (the
        single-process shard emulation of the tests) or one shard: unchanged."""
        sh = self._shard
        if sh is None or sh.world_size <= 1 or sh.local:
            return t                     # (a device group's shards live in ONE process: group.py copies shard 0's sequences)
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return t
        src = dist.get_global_rank(sh.group, 0) if sh.group is not None else 0
        if t.is_cuda and dist.get_backend(sh.group) == "gloo":
            h = t.detach().cpu().contiguous()           # test rigs (ranks sharing one GPU): via the host
            dist.broadcast(h, src=src, group=sh.group)
            return h.to(t.device)
        t = t.contiguous()
        dist.broadcast(t, src=src, group=sh.group)
        return t

    # ------------------------------------------------------------------------------------------
    # noise plumbing
    # ------------------------------------------------------------------------------------------
    def inject_noise(self, z):
        """Queue standard-normal draws in the reference's layout (K,T,nu) (KMPPI: (K,S,nu)) for
        the next `command()` instead of drawing them -- "identical inputs" for parity checks."""
        self._injected.append(z)

    def _noise_shape(self):
        return (self.K_local, self.T, self.nu)

    def _vec(self, t):
        """(nu,) parameter on device in dtype (0-dim bounds broadcast, mppi.py:124-126).  Cached on
        (tensor identity, in-place version) so that a steady-state command() launches no copy
        kernels for parameters, while assignments / in-place edits by the caller are picked up."""
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        key = (id(t), t._version, self.nu, self.dtype, str(self.d))
        hit = self._vec_cache.get(key)
        if hit is not None and hit[0] is t:
            return hit[1]
        v = t.detach().to(device=self.d, dtype=self.dtype)
        v = v.reshape(-1).expand(self.nu).contiguous() if v.numel() == 1 else v.reshape(-1).contiguous()
        if len(self._vec_cache) > 64:
            self._vec_cache.clear()
        self._vec_cache[key] = (t, v)
        return v

    def _static_key(self, Tn):
        """Identity + in-place version of everything the static part of the problem block is built
        from: a steady-state command() re-uses the cached block and parameter tensors, while
        attribute assignments / in-place edits by the caller (autotune, tests) are picked up."""
        # (this runs on every command: small problems are bound by the host's ~15 us per command, not by the device)
        m = self._model
        a, b, c, d, e, f = self.u_init, self.noise_mu, self._sigma_inv_kernel, self._noise_L, self.u_min, self.u_max
        T_ = torch.Tensor
        return (Tn, self.K_local, self.nx, self.nu, self.k_offset, id(m), m.hidden if m is not None else 0,
                bool(self.noise_abs_cost), bool(self.sample_null_action), int(self.u_per_command),
                float(self.lambda_), float(self.u_scale), int(self.M), float(self.rollout_var_cost),
                float(self.rollout_var_discount), self.seed,
                (id(a), a._version) if isinstance(a, T_) else a, (id(b), b._version) if isinstance(b, T_) else b,
                (id(c), c._version) if isinstance(c, T_) else c, (id(d), d._version) if isinstance(d, T_) else d,
                (id(e), e._version) if isinstance(e, T_) else e, (id(f), f._version) if isinstance(f, T_) else f,
                m._param_version if m is not None else 0)

    def _problem(self, Tn=None, U=None):
        """MppiProblem for this controller: static part cached (see _static_key), U bound fresh."""
        if self.d.type != "cuda":
            raise RuntimeError("pytorch_mppi_amd runs on the MI355X only: construct the controller with "
                               "device='cuda' (there is no CPU compute path)")
        Tn = Tn or self.T
        if getattr(self._model, "_param_tensors", None):
            self._model.refresh_params()           # a traced model's trainable tensors: re-gathered when written (jit.py)
        key = self._static_key(Tn)
        hit = self._problem_cache.get(Tn)
        if hit is None or hit[0] != key:
            p = N.MppiProblem()
            p.K, p.T, p.nx, p.nu = self.K_local, Tn, self.nx, self.nu
            p.noise_pitch = self._zpitch()
            p.S = 0
            p.dtype = _DT[self.dtype]
            p.k_offset = self.k_offset
            p.model_id = self._model.model_id if self._model is not None else N.MODEL_NONE
            p.hidden = self._model.hidden if self._model is not None else 0
            p.model_flags = self._model.flags() if self._model is not None else 0
            p.sigma_diagonal = int(self._diagonal_sigma)
            p.noise_abs_cost = int(bool(self.noise_abs_cost))
            p.sample_null_action = int(bool(self.sample_null_action))
            p.u_per_command = int(self.u_per_command)
            p.lambda_ = float(self.lambda_)
            p.u_scale = float(self.u_scale)
            p.rollout_samples = int(self.M)
            p.rollout_var_cost = float(self.rollout_var_cost)
            p.rollout_var_discount = float(self.rollout_var_discount)
            p.seed = self.seed
            p.philox_rounds = self.philox_rounds
            keep = dict(
                u_init=self._vec(self.u_init), mu=self._vec(self.noise_mu),
                L=self._noise_L.to(device=self.d, dtype=self.dtype).contiguous(),
                sinv=self._sigma_inv_kernel.to(device=self.d, dtype=self.dtype).contiguous(),
                umin=self._vec(self.u_min), umax=self._vec(self.u_max))
            p.u_init, p.noise_mu = _ptr(keep["u_init"]), _ptr(keep["mu"])
            p.noise_L, p.sigma_inv = _ptr(keep["L"]), _ptr(keep["sinv"])
            p.u_min, p.u_max = _ptr(keep["umin"]), _ptr(keep["umax"])
            if self._model is not None:
                keep["mp"] = self._model.param_blob(self.d, self.dtype)
                p.model_params = _ptr(keep["mp"])
                p.model_params_elems = int(keep["mp"].numel())
                if self._model.process_noise is not None:
                    keep["psd"] = self._model.process_noise.to(device=self.d, dtype=self.dtype).contiguous()
                    p.process_noise_sd = _ptr(keep["psd"])
            hit = (key, p, keep)
            self._problem_cache[Tn] = hit
        # a fresh struct per command (the previous one stays valid for the lazy attributes)
        p = N.MppiProblem.from_buffer_copy(hit[1])
        keep = dict(hit[2])
        Ut = self.U if U is None else U
        if Ut.device != self.d or Ut.dtype != self.dtype or not Ut.is_contiguous():
            Ut = Ut.to(device=self.d, dtype=self.dtype).contiguous()
        keep["U"] = Ut
        p.U = Ut.data_ptr()
        p._keep = keep      # keep the tensors alive as long as the struct
        return p

    def _zpitch(self):
        """Row pitch (samples) of this controller's TNK4 noise arrays (engine's choice, mppi_noise_pitch)."""
        key = (self.K_local, self.dtype)
        if getattr(self, "_zpitch_cache", (None, 0))[0] != key:
            self._zpitch_cache = (key, N.noise_pitch(self.K_local, _DT[self.dtype]))
        return self._zpitch_cache[1]

    def _zelems(self, Tn):
        """Elements of a TNK4 array for a (Tn, nu) sequence over this controller's samples."""
        return N.noise_rows4(Tn, self.nu) * self._zpitch() * 4

    def _row_buffer(self, n):
        """The TNK4 row array a command generates or converts its normals into.  ONE buffer per size,
        reused by every command (stream order makes that safe; the lazily materialised attributes only
        ever refer to the LAST command's rows): an allocation less per command."""
        buf = self._zbuf.get(n)
        if buf is None or buf.dtype != self.dtype:
            if len(self._zbuf) > 4:
                self._zbuf.clear()
            buf = self._zbuf[n] = torch.empty(n, device=self.d, dtype=self.dtype)
        return buf

    def _attach_workspace(self, p):
        key = (p.K, p.T, p.nu, p.num_envs)
        need = self._ws_need.get(key)
        if need is None:
            need = self._ws_need[key] = int(N.lib().mppi_workspace_elems(C.byref(p)))
        if self._ws is None or self._ws.numel() < need or self._ws.dtype != self.dtype:
            # zero-filled: the single-launch command keeps its arrival ticket in the last elements
            self._ws = torch.zeros(max(need, 1), device=self.d, dtype=self.dtype)
        p.workspace = self._ws.data_ptr()
        p.workspace_elems = self._ws.numel()

    def _stream(self):
        """The caller's current HIP stream (raw handle; torch.cuda.current_stream() costs ~10 us)."""
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self._dev_index))

    def _randn(self, *shape):
        # per-command sample draws; sharded torch modes draw from the shard's own generator
        return torch.randn(*shape, device=self.d, dtype=self.dtype, generator=self._shard_gen)

    def _draw_noise(self, p, shape):
        """Bind this command's standard normals to the problem: injected / torch.randn (reference
        layout, converted to the engine's sample-minor rows-of-4) or in-kernel Philox."""
        lib = N.lib()
        K, Tn, nu = shape
        self.last_draw = None
        if self.M > 1 and (self._injected or self.rng != "philox"):
            # the fused multi-rollout kernel keys its process-noise stream with the command number too
            self._call += 1
            p.call = self._call
        if self._injected:
            z = self._injected.pop(0)
            z = torch.as_tensor(z).to(device=self.d, dtype=self.dtype)
            if tuple(z.shape) == (self.K, Tn, nu) and self.K != K:
                z = z[self.k_offset:self.k_offset + K]          # global draw, this shard's rows
            if tuple(z.shape) != (K, Tn, nu):
                raise ValueError(f"injected noise has shape {tuple(z.shape)}, expected {(K, Tn, nu)}")
            z = z.contiguous()
        elif self.rng == "torch":
            if self.torch_rows and self._torch_stream_fill(p, K, Tn, nu):
                return                                                    # the same values, already in the engine's rows
            z = self._randn(K, Tn, nu)                                    # mppi.py:203
        elif self.rng == "torch-native":
            # same generator, drawn straight into the engine's sample-minor layout: no conversion
            # pass; which (k,t,n) gets which draw differs from the reference-layout draw
            zn = self._randn(self._zelems(Tn))
            p.noise_src = N.NOISE_TNK4
            p.z = _ptr(zn)
            p._keep["z"] = zn
            return
        else:
            self._call += 1
            p.noise_src = N.NOISE_PHILOX
            p.call = self._call
            p.z = None
            if self.philox_store and self._onchip_wanted(K, Tn, nu):
                self.last_draw = "philox-onchip"
                if self.onchip_spill:
                    # the rows that fit neither registers nor LDS wait for their sample's weight in this array (stored once,
                    # fetched once) instead of being generated a second time: 75.8 -> 71.3 us at C3 (include/mppi_amd.h, ABI 20)
                    key = (K, Tn, nu)
                    sp = self._spill if self._spill is not None and self._spill[0] == key else None
                    if sp is None:
                        n = int(lib.mppi_onchip_spill_elems(C.byref(p)))
                        sp = self._spill = (key, torch.empty(n, device=self.d, dtype=self.dtype) if n > 0 else None)
                    if sp[1] is not None:
                        p.onchip_spill, p.onchip_spill_elems = _ptr(sp[1]), sp[1].numel()
                        p._keep["spill"] = sp[1]
                return
            if self.philox_store:
                # generate once, keep the rows for K3 to re-read: Philox + Box-Muller costs more per
                # element than an HBM read
                rows4 = N.noise_rows4(Tn, nu)
                n = self._zelems(Tn)
                # inside K1 every lane generates its own rows one after the other (~0.35 us per
                # row-of-4, however small K is); the generator launch spreads them over the whole chip
                # and costs one launch (~4 us): it wins from ~16 rows per sample on (tools/k_sweep.py)
                fill = self.philox_fill if self.philox_fill is not None else rows4 >= 16
                if self.M > 1 and not self._needs_generic():
                    fill = True            # the multi-rollout K1 reads its rows from memory
                self.last_draw = "philox-fill" if fill else "philox-k1"
                pf, self._pf_rows = self._pf_rows, None
                self._next_armed = None
                ahead = (self.draw_ahead and self.draw_ahead_philox and self.dtype == torch.float32 and (self._diagonal_sigma or not self.coloured_fill)
                         and self.d.type == "cuda" and not self._in_capture and K * Tn * nu >= self.draw_ahead_min)
                if pf is not None and pf[0] == (K, Tn, nu, int(p.k_offset), int(p.seed), int(p.call)) and (fill or pf[2]):
                    # the rows of THIS command exist already: generated inside the previous command's K3 launch (ABI 21,
                    # csrc/noise_torch.hip -- the VALU that HBM-bound launch leaves idle; for small commands, the CUs) or while the
                    # previous command's collective ran.  Rows are a pure function of (seed, command, sample, row).
                    zn = pf[1]
                    self._pf_hits += 1
                    if pf[2]:
                        self._zbuf_alt[n], self._zbuf[n] = self._zbuf.get(n), zn        # the two row buffers change roles
                        self.last_draw = "philox-rows-ahead"
                    p.z = _ptr(zn)
                    p._keep["z"] = zn
                    p.noise_src = N.NOISE_TNK4
                    if ahead:
                        self._arm_next_philox(p, n)
                    return
                zn = self._row_buffer(n)
                p.z = _ptr(zn)
                p._keep["z"] = zn
                if ahead:
                    self._arm_next_philox(p, n)
                if fill:
                    # a separate generator launch at full occupancy (32 us for C3's 50 M normals, write
                    # floor 26 us), then K1 as the pure HBM-read kernel.  Short horizons keep the
                    # generation inside K1: one launch fewer.
                    if not self._diagonal_sigma and self.coloured_fill:
                        # full Sigma: the generator applies chol(Sigma) z + mu itself (full occupancy,
                        # a few us) and K1 / K3 run their diagonal form on the coloured rows instead of
                        # doing nu*(nu+1)/2 FMAs per timestep behind LDS reads at one wave per SIMD
                        rc = lib.mppi_noise_fill_philox_coloured(C.byref(p), p.z, self._stream())
                        if rc == 0:
                            p.noise_src, p.noise_coloured = N.NOISE_TNK4, 1
                            return
                        if rc != N.E_UNSUPPORTED:
                            N.check(rc, "mppi_noise_fill_philox_coloured")
                    N.check(lib.mppi_noise_fill_philox(C.byref(p), p.z, self._stream()), "mppi_noise_fill_philox")
                    p.noise_src = N.NOISE_TNK4
            return
        p._keep["z_ktn"] = z
        if self._ktn_direct_ok(p, Tn, nu, z):
            # fused fp32 path, diagonal Sigma: K1 and K3 read the reference-layout draw in place
            p.noise_src = N.NOISE_KTN
            p.z = _ptr(z)
            return
        self._convert_noise(p)

    def _torch_stream_fill(self, p, K, Tn, nu):
        """rng="torch": the values `torch.randn(K, Tn, nu)` would produce from the generator's present state, written by
        the engine's own launch straight into the rows K1 / K3 stream (csrc/noise_torch.hip, `mppi_noise_fill_torch`),
        and the generator advanced exactly as that call advances it -- every draw of the process, before and after, is
        what it would have been.  The first draw of every shape is compared with torch.randn itself, bit for bit, and
        the generator's offset with ATen's rule; a disagreement (another torch, another rocrand) switches this off for
        the process and the command draws with torch.randn as before.  False: not applicable here."""
        if (self.dtype != torch.float32 or (Tn * nu) % 4 or self.d.type != "cuda" or self._in_capture
                or _TORCH_ROWS.get("off") or torch.cuda.is_current_stream_capturing()):
            # (a capture: torch.randn registers its generator with the graph and replays advance it; the offset this launch
            # takes as an argument would be frozen -- capture_command() says so itself, a user's own torch.cuda.graph() is
            # caught by the query)
            return False
        gen = self._shard_gen if self._shard_gen is not None else torch.cuda.default_generators[self._dev_index]
        numel = K * Tn * nu
        cap = _TORCH_ROWS.get(("cap", self._dev_index))
        if cap is None:
            props = torch.cuda.get_device_properties(self._dev_index)
            cap = _TORCH_ROWS[("cap", self._dev_index)] = props.multi_processor_count * (props.max_threads_per_multi_processor // 256)
        grid = min(cap, (numel + 255) // 256)
        inc = ((numel - 1) // (1024 * grid) + 1) * 4
        lib = N.lib()
        zn = self._row_buffer(self._zelems(Tn))
        pitch = self._zpitch()
        key = (self._dev_index, K, Tn, nu)
        if key not in _TORCH_ROWS:
            # once per shape and process: is this what torch.randn does here?
            state = gen.get_state()
            seed, off = gen.initial_seed(), gen.get_offset()
            ref = torch.randn(K, Tn, nu, device=self.d, dtype=self.dtype, generator=gen)
            moved = gen.get_offset() - off
            gen.set_state(state)
            rc = lib.mppi_noise_fill_torch(_ptr(zn), K, Tn, nu, pitch, seed, off, grid, self._stream())
            if rc == N.E_UNSUPPORTED:
                _TORCH_ROWS[key] = False           # a shape the launch does not take (more than 65535 rows-of-4): torch.randn
                return False
            ok = rc == 0 and moved == inc
            if ok:
                rows = zn.view(-1, pitch, 4)[:, :K, :].permute(1, 0, 2).reshape(K, Tn, nu)
                ok = torch.equal(rows, ref)
            _TORCH_ROWS[key] = ok
            if not ok:
                import logging
                _TORCH_ROWS["off"] = True
                logging.getLogger("pytorch_mppi_amd").warning(
                    "pytorch_mppi_amd: torch.randn(%d, %d, %d) is not the stream csrc/noise_torch.hip reproduces (rc %d, generator "
                    "offset +%d against +%d expected): rng='torch' keeps drawing with torch.randn", K, Tn, nu, rc, moved, inc)
                return False
        elif not _TORCH_ROWS[key]:
            return False
        off, seed = gen.get_offset(), gen.initial_seed()
        nd, self._next_draw, self._next_armed = self._next_draw, None, None
        nkey = (K, Tn, nu, pitch, grid)
        if nd is not None and nd[0] == nkey and nd[1] is gen and nd[2] == seed and nd[3] == off:
            # command n-1's K3 launch generated exactly this draw beside its row stream (ABI 21, csrc/noise_torch.hip): the
            # generator is where that launch assumed it would be -- same seed, same offset: the same values, by construction.
            # The two row buffers change roles
            n_el = self._zelems(Tn)
            self._zbuf_alt[n_el], self._zbuf[n_el] = zn, nd[4]
            zn = nd[4]
            self._next_hits += 1
            self._next_misses = 0
            self.last_draw = "torch-rows-ahead"
        else:
            if nd is not None:
                self._next_misses += 1        # generated for nothing: somebody else drew from the generator (reset(), the user's own randn)
            N.check(lib.mppi_noise_fill_torch(_ptr(zn), K, Tn, nu, pitch, seed, off, grid, self._stream()), "mppi_noise_fill_torch")
            self.last_draw = "torch-rows"
        gen.set_offset(off + inc)
        p.noise_src = N.NOISE_TNK4
        p.z = _ptr(zn)
        p._keep["z"] = zn
        self._next_cmds += 1
        if self.draw_ahead and numel >= self.draw_ahead_min and (self._next_misses < 2 or self._next_cmds % 64 == 0):
            # (a caller that draws from the generator between every two commands -- the reference's benchmark protocol calls
            # reset() -- makes every draw-ahead useless and K3 pays for it: after two misses in a row it is tried only every 64th
            # command)
            # ... and this command's K3 launch generates the NEXT draw -- the values torch.randn will produce from (seed,
            # off + inc) if nobody else draws from this generator in between -- into the other row buffer, on the VALU the
            # HBM-bound row stream leaves idle.  Whether the engine did (only the streaming diagonal K3 carries it) is read
            # back behind the command (_settle_next); whether the assumption held is checked above, at the next command
            n_el = self._zelems(Tn)
            alt = self._zbuf_alt.get(n_el)
            if alt is None or alt.dtype != self.dtype or alt.device != zn.device:
                if len(self._zbuf_alt) > 2:
                    self._zbuf_alt.clear()
                alt = self._zbuf_alt[n_el] = torch.empty(n_el, device=self.d, dtype=self.dtype)
            p.next_z, p.next_seed, p.next_philox_offset, p.next_grid_blocks, p.next_kind = _ptr(alt), seed, off + inc, grid, N.NEXT_DRAW_TORCH
            p._keep["next_z"] = alt
            self._next_armed = (nkey, gen, seed, off + inc, alt)
        return True

    def _arm_next_philox(self, p, n_el):
        """rng="philox", rows in memory: let this command's K3 launch generate the rows of the NEXT command (call + 1) into the
        other row buffer (MppiProblem.next_*, kind MPPI_NEXT_DRAW_PHILOX); _settle_next reads back whether it did"""
        alt = self._zbuf_alt.get(n_el)
        if alt is None or alt.dtype != self.dtype or alt.device != self.d or alt.data_ptr() == p.z:
            if len(self._zbuf_alt) > 2:
                self._zbuf_alt.clear()
            alt = self._zbuf_alt[n_el] = torch.empty(n_el, device=self.d, dtype=self.dtype)
        p.next_z, p.next_seed, p.next_philox_offset, p.next_grid_blocks, p.next_kind = _ptr(alt), int(p.seed), int(p.call) + 1, 0, N.NEXT_DRAW_PHILOX
        p._keep["next_z"] = alt
        self._next_armed = ("philox", (int(p.K), int(p.T), int(p.nu), int(p.k_offset), int(p.seed), int(p.call) + 1), alt)

    def _settle_next(self, took=None):
        """behind the launches of a command: did its K3 generate the next draw (mppi_last_next_draw, thread-local -- `took`:
        what the thread that issued the launches read there)?"""
        armed, self._next_armed = self._next_armed, None
        if armed is not None and (int(N.lib().mppi_last_next_draw()) if took is None else int(took)) == 1:
            if armed[0] == "philox":
                self._pf_rows = (armed[1], armed[2], True)
            else:
                self._next_draw = armed

    def _onchip_wanted(self, K, Tn, nu):
        """rng="philox": does this command go without a row array (include/mppi_amd.h, ABI 18; scope as checked again by
        the engine, csrc/rollout_onchip.hpp `onchip_problem_ok`)?"""
        if self.philox_onchip is False or self._onchip_refused:
            return False
        ok = (type(self).__name__ in ("MPPI", "SMPPI") and self.dtype == torch.float32 and self.M == 1
              and self.specific_action_sampler is None and Tn == self.T and not self._needs_generic()
              and self._model.model_id != N.MODEL_MLP)      # the dense MLP has its own matrix-core K1
        if not ok:
            return False
        if not self._diagonal_sigma:
            # a full Sigma CAN run on chip (L z + mu per timestep in the lane; csrc/rollout_onchip.hpp behind
            # MPPI_ONCHIP_FULL_SIGMA, tested at full size), but the factor rows come out of LDS every timestep and the
            # kernel becomes LDS-issue-bound: 0.127 ms at C3 against 0.104 ms for rows coloured by the generator launch
            # and streamed (profiles/r03_variants_philox.txt) -- not in the product build
            return False
        if self.philox_onchip:
            return True
        # On chip every lane generates its own rows one after the other (~0.35 us per row-of-4 however small K is): the
        # launch costs the same ~80 us at C3's horizon for K = 1024 and K = 65536, while the streaming form spreads the
        # generation over the chip.  Measured at T = 64, nu = 12 (tools/k_sweep.py, profiles/r03_k_sweep.txt against
        # r02_k_sweep.txt): K = 16384 0.083 vs 0.056 ms, K = 65536 0.087 vs 0.106, K >= 262144 8.0e8 vs 5.9e8 rollouts/s ->
        # from three quarters of a full chip (one wave per SIMD = 65536 samples) upwards
        return K >= 49152

    def _ktn_direct_ok(self, p, Tn, nu, z):
        return (self.ktn_direct and self.M == 1 and self.dtype == torch.float32 and self._diagonal_sigma and (Tn * nu) % 4 == 0
                and nu in (4, 8, 12, 16) and z.data_ptr() % 16 == 0 and p.num_envs <= 1 and Tn == self.T
                and not self._needs_generic())

    def _convert_noise(self, p):
        """(K,T,nu) draw kept in p._keep['z_ktn'] -> the engine's sample-minor rows-of-4."""
        z = p._keep["z_ktn"]
        K, Tn, nu = z.shape
        zn = self._row_buffer(self._zelems(Tn))
        N.check(N.lib().mppi_noise_from_ktn(C.byref(p), _ptr(z), _ptr(zn), self._stream()), "mppi_noise_from_ktn")
        p.noise_src = N.NOISE_TNK4
        p.z = _ptr(zn)
        p._keep["z"] = zn

    # ------------------------------------------------------------------------------------------
    # command
    # ------------------------------------------------------------------------------------------
    def command(self, state, shift_nominal_trajectory=True, info=None):
        """mppi.py:240-252: returns the (nu,) / (u_per_command,nu) action as a device tensor,
        without synchronising."""
        self.info = info
        if self._jit_pending is not None:
            self._adopt_background_model()         # only HERE, between two commands: a command never changes path half-way
        if getattr(self._model, "watch", None) is not None:
            self._check_traced(state)              # traced callables: do they still say what the functor computes?
        return self._command(state, bool(shift_nominal_trajectory))

    def capture_command(self, state, shift_nominal_trajectory=True, warmup=3):
        """Capture one `command()` -- noise draw, K1, K3, K4 and the U hand-over -- into a HIP graph
        and return a `GraphedCommand`; replaying it costs one graph launch instead of ~6 kernel
        launches + ~30 us of host work (launch-bound configs such as 8192 x 32 gain ~3x).
        Constraints: fused or generic path with capturable callbacks; rng 'torch' / 'torch-native'
        (torch's generator advances correctly under graph replay; the Philox call counter is a
        launch argument and would be frozen); single shard; parameters (lambda_, bounds, ...) are
        frozen at capture -- capture again after changing them."""
        if type(self) is not MPPI:
            # SMPPI / KMPPI re-bind `action_sequence` / `theta` to fresh tensors every command; a captured
            # graph would keep replaying the capture-time pointers
            raise NotImplementedError(f"capture_command supports plain MPPI only, not {type(self).__name__}")
        if self.rng == "philox":
            raise ValueError("capture_command needs rng='torch' or 'torch-native' (see docstring)")
        if self._sharded():
            raise ValueError("capture_command is single-shard")
        return GraphedCommand(self, state, bool(shift_nominal_trajectory), warmup)

    def _host_state_to_device(self, state):
        """A small host-resident state (what a simulator hands back every step) travels inside the launch packet
        of a one-wave kernel (`mppi_upload_small`): ~5 us of host time, no staging buffer to keep alive.  The
        pageable `.to(device)` stalls the host for ~20 us per command -- the whole budget of a small problem."""
        src = state.detach().to(dtype=self.dtype).contiguous()            # host-side cast (a no-op for matching dtypes)
        out = torch.empty(src.shape, dtype=self.dtype, device=self.d)
        N.check(N.lib().mppi_upload_small(src.data_ptr(), src.numel() * src.element_size(), _ptr(out), self._stream()),
                "mppi_upload_small")
        return out

    def _to_state(self, state):
        if not torch.is_tensor(state):
            state = torch.as_tensor(state)
        if state.device.type == "cpu" and self.d.type == "cuda" and 0 < state.numel() * (8 if self.dtype == torch.float64 else 4) <= 2048:
            state = self._host_state_to_device(state)
        else:
            state = state.to(dtype=self.dtype, device=self.d)             # mppi.py:262-264
        if self.K_local != self.K and tuple(state.shape) == (self.K, self.nx):
            # per-sample initial states (mppi.py:302) of the GLOBAL problem: this shard's rows
            state = state[self.k_offset:self.k_offset + self.K_local]
        return state

    def _sampler_rows(self, p):
        """mppi.py:393-399: rows [null, null+n) come from the sampler; global indices."""
        s = self.specific_action_sampler
        if s is None:
            return
        actions = s.sample_trajectories(self.state, self.info)
        actions = torch.as_tensor(actions).to(device=self.d, dtype=self.dtype).reshape(-1, self.T, self.nu).contiguous()
        i = 1 if self.sample_null_action else 0
        s.register_sample_start_end(i, i + actions.shape[0])
        p.n_sampler_rows = actions.shape[0]
        p.sampler_actions = _ptr(actions)
        p._keep["sampler"] = actions

    def _adopt_background_model(self):
        """the background compile of the traced callables (auto_jit="async") has finished: switch to the fused kernels"""
        th, box = self._jit_pending
        if th.is_alive():
            return
        self._jit_pending = None
        import logging
        log = logging.getLogger("pytorch_mppi_amd")
        m = box.get("model")
        if m is None:
            self.jit_note = f"generic path: {type(box.get('error')).__name__}: {box.get('error')}"
            log.warning("pytorch_mppi_amd: dynamics / running_cost stay on the generic (callback) path: %s", self.jit_note)
            return
        m.watch = box["watch"]
        if not self._settle_watch(m):
            # the callables' state moved again while these kernels were being compiled: they are already out of date
            self._jit_retraces += 1
            if self._jit_retraces <= 16:
                for path in m.watch.tensors_at(m.watch.changed()):
                    if not any(q.holder is path.holder and q.key == path.key for q in self._jit_dynamic):
                        self._jit_dynamic.append(path)
                dyn, rc, term, sd = self._callables()
                self._model = self._try_trace(dyn, rc, term, sd, background=True, dynamic=self._jit_dynamic, verify_in_background=True)
            return
        self._model = m
        self._jit_cmds = 0
        self._jit_next_check, self._jit_last_check = 0, None
        self._problem_cache.clear()
        self.jit_note = f"fused: traced {m.traced_ops} operations per sample into {m.name} (compiled in the background)"
        log.warning("pytorch_mppi_amd: %s", self.jit_note)

    def wait_for_jit(self, timeout=None):
        """Block until a background compile (auto_jit="async") has finished; True when the controller runs fused afterwards."""
        if self._jit_pending is not None:
            self._jit_pending[0].join(timeout)
            self._adopt_background_model()
        return self._model is not None

    def _needs_generic(self):
        # (twice per command, and the host's ~15 us per command are what bounds a small problem: memoised on what it reads)
        m = self._model
        if m is None:
            return True
        s = self.specific_action_sampler
        key = (id(m), m.model_id, self.M, id(s), id(m.process_noise), self.nx, self.nu, self.dtype)
        hit = self._generic_memo
        if hit is not None and hit[0] == key:
            return hit[1]
        v = self._needs_generic_now()
        self._generic_memo = (key, v)
        return v

    def _needs_generic_now(self):
        if self._model is None:
            return True
        if self.M != 1 and not self._fused_multi_ok():
            return True
        if self._model.process_noise is not None and not (self.M > 1 and self._fused_multi_ok()):
            # a model with process noise is stochastic whatever M is (models.NativeModel.with_process_noise); only the
            # fused multi-rollout kernel draws it on the device -- everything else keeps the callables' own noise
            return True
        s = self.specific_action_sampler
        if s is not None and type(s).specific_dynamics is not SpecificActionSampler.specific_dynamics:
            return True      # arbitrary Python post-processing of the dynamics (mppi.py:315-317)
        p_ok = N.model_supported(self._model.model_id, self.nx, self.nu, _DT[self.dtype], self._model.hidden)
        return not p_ok

    def _fused_multi_ok(self):
        """M > 1 rollouts per action sequence inside K1 (csrc/rollout.hpp rollout_stream_multi): MPPI, SMPPI and KMPPI (its
        two-launch form: interpolated raw actions in memory), at most 4 copies of the state per lane; anything else runs
        the reference's callback loop."""
        return (1 < self.M <= 4 and self.specific_action_sampler is None and not getattr(self._model, "heavy", False))

    def _command(self, state, shift):
        p = self._begin(state, shift)
        if self._sharded() and not getattr(p, "_combined", False):
            if getattr(self._shard, "local", False):
                raise RuntimeError("this controller is one shard of a device group (MPPI(..., devices=[...])): command the group, not the shard")
            comm = None if self.overlap_collective else self._shard.native_comm(self.d)
            if comm is not None:
                # generic path: the engine issues the record all-gather itself (RCCL C API on this stream) + K5
                self._exchange_native(p, comm)
                return self._end(p)
            if self.overlap_collective and self.last_draw == "philox-fill" and not self._injected:
                records, work = self._shard.all_gather_start(p._keep["record"])
                self._prefetch_philox_rows(p)   # queued behind K4, runs while the collective is in flight
                if work is not None:
                    work.wait()
            else:
                records = self._shard.all_gather(p._keep["record"])
            self._combine(p, records)
        return self._end(p)

    def _exchange_native(self, p, comm):
        records = torch.empty(comm.world_size, 2 + p.T * p.nu, device=self.d, dtype=self.dtype)
        p._keep["records"] = records
        N.check(N.lib().mppi_exchange_combine(C.byref(p), comm.handle, _ptr(records), comm.world_size, self._stream()),
                "mppi_exchange_combine")

    def _prefetch_philox_rows(self, p):
        """Sharded commands: the Philox rows of the NEXT command are a pure function of
        (seed, call+1, sample, row) -- nothing of this command's result enters -- so their generator
        launch is queued before the caller's stream waits for the record all-gather: the
        latency-bound collective (tens of microseconds over xGMI) hides behind 30 us of generation.
        The next command picks the buffer up if (shape, seed, call) still match, else drops it."""
        if p.noise_coloured:
            self._pf_rows = None
            return
        q = N.MppiProblem.from_buffer_copy(p)
        q.call = self._call + 1
        q.noise_src = N.NOISE_PHILOX
        n = self._zelems(q.T)
        zn = torch.empty(n, device=self.d, dtype=self.dtype)
        N.check(N.lib().mppi_noise_fill_philox(C.byref(q), _ptr(zn), self._stream()), "mppi_noise_fill_philox")
        self._pf_rows = ((q.K, q.T, q.nu, int(q.k_offset), int(q.seed), int(q.call)), zn, False)

    def _sharded(self):
        # _force_collective: measurement seam (tools/shard_overhead.py) -- run record -> all_gather -> K5 at world_size 1
        return self._shard is not None and (self._shard.world_size > 1 or self._force_collective)

    def _begin(self, state, shift):
        """Everything local to this shard: noise, K1 (or the generic callback loop), K3, K4.
        Single shard: K4 also applies the update.  Sharded: K4 only writes the shard record."""
        p = self._prepare(state, shift)
        if p._deferred:
            self._launch_prepared(p)
        return p

    def _prepare(self, state, shift):
        """The host part of a command up to (not including) the fused path's launch: the problem block with this command's
        draw, buffers and state bound.  `p._deferred`: the fused launch is still to be issued -- by `_launch_prepared` on
        this thread, or by the device group's worker thread of this shard's device (group.py, csrc/group.hip).  The generic
        (callback) path cannot be handed over: it has run K1's stand-in, K3 and K4 when this returns (`_deferred` False)."""
        lib = N.lib()
        self.state = self._to_state(state)
        p = self._problem()
        p.shift = int(shift)
        st = self._stream()
        self._attach_workspace(p)
        self._draw_noise(p, self._noise_shape())
        self._sampler_rows(p)
        K = self.K_local
        cost_total = torch.empty(K, device=self.d, dtype=self.dtype)
        p.cost_total = _ptr(cost_total)
        per_sample = tuple(self.state.shape) == (K, self.nx)              # mppi.py:302
        self._states = self._actions = self._noise = self._perturbed_action = None

        apply = 0 if self._sharded() else 1
        # omega = (1/eta) exp(-(c - beta)/lambda) and cost_total_non_zero (mppi.py:256-258) are functions of
        # cost_total and the record {beta, eta, ...}: a single-shard command leaves them to their first
        # read (two allocations and a pass over K less per command, and what lets a small problem run
        # as ONE launch); a sharded one has K5 rescale them, so there they are written
        lazy = apply == 1
        omega = None if lazy else torch.empty(K, device=self.d, dtype=self.dtype)
        wnz = None if lazy else torch.empty(K, device=self.d, dtype=self.dtype)
        U_new = torch.empty(self.T, self.nu, device=self.d, dtype=self.dtype)
        if lazy:
            # single shard: the record {beta, eta, P} is only read back by the lazily derived weights of
            # THIS command -> one buffer for all commands
            record = self._rec_buf
            if record is None or record.numel() != 2 + self.T * self.nu or record.dtype != self.dtype:
                record = self._rec_buf = torch.empty(2 + self.T * self.nu, device=self.d, dtype=self.dtype)
        else:
            record = torch.empty(2 + self.T * self.nu, device=self.d, dtype=self.dtype)
        p.omega, p.cost_total_non_zero, p.U_out, p.record = _ptr(omega), _ptr(wnz), _ptr(U_new), _ptr(record)
        p._keep.update(omega=omega, wnz=wnz, U_new=U_new, record=record)
        self._lazy_w = (float(self.lambda_), record) if lazy else None
        self.cost_total = cost_total

        if not self._needs_generic():
            s0 = self._fused_state(per_sample)
            p.state = _ptr(s0)
            p._keep["state"] = s0
            p.state_per_sample = int(per_sample)
            p.use_terminal = int(self.terminal_state_cost is not None)
            p._deferred, p._apply = True, apply
            return p

        p._deferred = False
        self._generic_total_cost(p, cost_total, st)
        if p.noise_src == N.NOISE_PHILOX and p.z:
            p.noise_src = N.NOISE_TNK4            # the rows mppi_prepare generated are in p.z now
        N.check(lib.mppi_weights_partial(C.byref(p), st), "mppi_weights_partial")
        self._settle_next()
        N.check(lib.mppi_finalize(C.byref(p), apply, st), "mppi_finalize")
        return p

    def _launch_prepared(self, p):
        """the fused path's launches of a prepared command, on the calling thread: K1 + K3 + K4 from one C call (a sharded rank
        with an engine-owned communicator: + the record all-gather + K5)"""
        lib, st, apply = N.lib(), self._stream(), p._apply
        comm = None
        if apply == 0 and not self.overlap_collective and not getattr(self._shard, "local", False):
            comm = self._shard.native_comm(self.d)

        def launch():
            if comm is None:
                return lib.mppi_command(C.byref(p), apply, st)            # K1 + K3 + K4, one call
            # sharded: K1 + K3 + K4 + ncclAllGather + K5 on this stream, one call
            records = torch.empty(comm.world_size, 2 + p.T * p.nu, device=self.d, dtype=self.dtype)
            p._keep["records"] = records
            p._combined = True
            return lib.mppi_command_sharded(C.byref(p), comm.handle, _ptr(records), comm.world_size, st)

        rc = launch()
        if rc == N.E_UNSUPPORTED and p.noise_src == N.NOISE_KTN:
            self.ktn_direct = False            # no in-place instantiation for this model: convert from now on
            self._convert_noise(p)
            rc = launch()
        N.check(rc, "mppi_command")
        self._launched(p, int(lib.mppi_last_command_form()), int(lib.mppi_last_next_draw()))

    def _group_blocks(self, p):
        """what a device group's worker issues for this prepared command (csrc/group.hip): (problem, theta problem | None)"""
        return p, None

    def _launched(self, p, form, next_draw):
        """behind the fused launches of a command (issued here or by the device group's worker): what the engine reported"""
        p._deferred = False
        self._settle_next(next_draw)
        if self.last_draw == "philox-onchip" and form != N.FORM_ONCHIP:
            # the engine ran K1 + K3 with the rows generated twice instead (a model without the on-chip kernel, ...):
            # correct, slower -- store the rows from the next command on
            self._onchip_refused = True
            self.last_draw = "philox-twice"
        if p.noise_src == N.NOISE_PHILOX and p.z:
            p.noise_src = N.NOISE_TNK4        # the rows K1 generated are in p.z now (lazy attributes)

    def _fused_state(self, per_sample):
        """Initial state as the fused kernels read it: (K_local,nx) rows or one (nx,) vector.  The
        reference expands anything else to (K, numel) and lets the callbacks cope (mppi.py:305); a
        compiled model has exactly nx state registers, so other sizes are refused."""
        if per_sample:
            return self.state.contiguous()
        if self.state.numel() != self.nx:
            raise ValueError(f"state has shape {tuple(self.state.shape)}; the fused path takes (nx,) = ({self.nx},) "
                             f"or per-sample ({self.K}, {self.nx})")
        return self.state.reshape(-1).contiguous()

    def _combine(self, p, records):
        """K5: identical rank-order combination of the all-gathered shard records on every rank."""
        p._keep["records"] = records
        N.check(N.lib().mppi_combine(C.byref(p), _ptr(records), int(records.shape[0]), self._stream()),
                "mppi_combine")

    def _end(self, p):
        self._omega = p._keep["omega"]
        self._wnz = p._keep["wnz"]
        self._record = p._keep["record"]
        self._last = p                # keeps z / U / sampler tensors alive for the lazy attributes
        self.U = p._keep["U_new"]                                         # mppi.py:270 (new tensor)
        if self.u_per_command == 1:
            return self.U[0]                                              # :271-275 (one view instead of two)
        return self.U[:self.u_per_command]

    # ------------------------------------------------------------------------------------------
    # generic (callback) path: mppi.py:297-332 around the engine's prepare kernel
    # ------------------------------------------------------------------------------------------
    def _generic_total_cost(self, p, cost_total, st):
        lib = N.lib()
        K, T, nu = self.K_local, self.T, self.nu
        pa = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        noise = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        pert = torch.empty(K, device=self.d, dtype=self.dtype)
        p.perturbed_action, p.noise, p.pert_cost = _ptr(pa), _ptr(noise), _ptr(pert)
        N.check(lib.mppi_prepare(C.byref(p), st), "mppi_prepare")
        p.perturbed_action = p.noise = p.pert_cost = None
        self._perturbed_action, self._noise = pa, noise
        # (no autograd graph through the callbacks: a dynamics network with trainable parameters -- the reference's
        # tests/pendulum_approximate.py -- would otherwise drag requires_grad into cost_total and keep T steps of history)
        with torch.no_grad():
            rollout_cost, self._states, actions = self._compute_rollout_costs(pa)
        self._actions = actions / self.u_scale if actions is not None else None
        torch.add(rollout_cost, pert, out=cost_total)                     # mppi.py:416
        N.check(lib.mppi_cost_block_min(C.byref(p), st), "mppi_cost_block_min")

    def _compute_rollout_costs_multi(self, perturbed_actions):
        """M > 1 state rollouts per action sequence with the discounted variance cost, as
        mppi.py:334-373 (callbacks see M*K rows); generic path only."""
        K, T, nu = perturbed_actions.shape
        M = self.M
        cost_samples = torch.zeros(M, K, device=self.d, dtype=self.dtype)
        cost_var = torch.zeros(K, device=self.d, dtype=self.dtype)
        if tuple(self.state.shape) == (K, self.nx):
            state0 = self.state
        else:
            state0 = self.state.view(1, -1).expand(K, -1)
        state0 = state0.repeat(M, 1, 1)
        states = torch.empty(M, K, T, self.nx, device=self.d, dtype=self.dtype)
        actions = torch.empty(M, K, T, nu, device=self.d, dtype=self.dtype)
        flat = state0.reshape(M * K, self.nx)
        sampler = self.specific_action_sampler
        for t in range(T):
            u = self.u_scale * perturbed_actions[:, t].expand(M, -1, -1)
            flat = self._dynamics_fn(flat, u.reshape(M * K, nu), t)
            if sampler is not None:
                flat = sampler.specific_dynamics(flat.reshape(M, K, -1), state0.reshape(M, K, -1), u, t).reshape(M * K, -1)
            c = self._running_cost_fn(flat, u.reshape(M * K, nu), t).reshape(M, K)
            cost_samples = cost_samples + c
            cost_var += c.var(dim=0) * self._var_discount_factors[t]
            states[:, :, t] = flat.reshape(M, K, -1)[:, :, :self.nx]
            actions[:, :, t] = u
        cost_samples = cost_samples + self._terminal_state_cost_fn(states, actions)
        cost_total = cost_samples.mean(dim=0) + cost_var * self.rollout_var_cost
        return cost_total, states, actions

    def _compute_rollout_costs(self, perturbed_actions):
        """The user-callback T-loop, as mppi.py:297-332 (M == 1)."""
        if self.M > 1:
            return self._compute_rollout_costs_multi(perturbed_actions)
        K, T, nu = perturbed_actions.shape
        cost_total = torch.zeros(K, device=self.d, dtype=self.dtype)
        if tuple(self.state.shape) == (K, self.nx):
            state = self.state.clone()
        else:
            state = self.state.view(1, -1).expand(K, -1)
        need_storage = self.terminal_state_cost is not None
        if need_storage:
            states = torch.empty(1, K, T, self.nx, device=self.d, dtype=self.dtype)
            actions = torch.empty(1, K, T, nu, device=self.d, dtype=self.dtype)
        sampler = self.specific_action_sampler
        for t in range(T):
            u = self.u_scale * perturbed_actions[:, t]
            state = self._dynamics_fn(state, u, t)
            if sampler is not None:
                state = sampler.specific_dynamics(state.unsqueeze(0), state.unsqueeze(0), u.unsqueeze(0), t).squeeze(0)
            c = self._running_cost_fn(state, u, t)
            cost_total = cost_total + c.reshape(K)
            if need_storage:
                states[0, :, t] = state[:, :self.nx]
                actions[0, :, t] = u
        if need_storage:
            c = self._terminal_state_cost_fn(states, actions)
            if torch.is_tensor(c) and c.dim() > 1:
                c = c.squeeze(0)
            cost_total = cost_total + c
        else:
            states = actions = None
        return cost_total, states, actions

    # ------------------------------------------------------------------------------------------
    # lazily materialised public attributes of the fused path (mppi.py:383-385, :411-412)
    # ------------------------------------------------------------------------------------------
    def _materialize(self):
        if self._last is None:
            return
        lib = N.lib()
        p = self._last
        if p.noise_src == N.NOISE_KTN:
            self._convert_noise(p)
        K, T, nu = self.K_local, self.T, self.nu
        pa = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        noise = torch.empty(K, T, nu, device=self.d, dtype=self.dtype)
        p.perturbed_action, p.noise = _ptr(pa), _ptr(noise)
        N.check(lib.mppi_prepare(C.byref(p), self._stream()), "mppi_prepare")
        p.perturbed_action = p.noise = None
        self._perturbed_action, self._noise = pa, noise

    def _derive_weights(self):
        lam, record = self._lazy_w
        w = torch.exp((-1.0 / lam) * (self.cost_total - record[0]))     # mppi.py:12-13, :256 (beta = record[0])
        self._wnz = w
        self._omega = (1.0 / record[1]) * w                             # :257-258 (eta = record[1])

    @property
    def omega(self):
        if self._omega is None and self._lazy_w is not None and self.cost_total is not None:
            self._derive_weights()
        return self._omega

    @omega.setter
    def omega(self, v):
        self._omega = v
        self._lazy_w = None if v is None else self._lazy_w

    @property
    def cost_total_non_zero(self):
        if self._wnz is None and self._lazy_w is not None and self.cost_total is not None:
            self._derive_weights()
        return self._wnz

    @cost_total_non_zero.setter
    def cost_total_non_zero(self, v):
        self._wnz = v

    @property
    def noise(self):
        if self._noise is None:
            self._materialize()
        return self._noise

    @noise.setter
    def noise(self, v):
        self._noise = v

    @property
    def perturbed_action(self):
        if self._perturbed_action is None:
            self._materialize()
        return self._perturbed_action

    @perturbed_action.setter
    def perturbed_action(self, v):
        self._perturbed_action = v

    @property
    def states(self):
        """Visited states: (1,K,T,nx), like the reference only kept when a terminal cost is set
        (mppi.py:307-310, :329-331) -- or (M,K,T,nx) for M > 1 rollouts, where the reference always
        stores them (:349-350, :366)."""
        want = self.terminal_state_cost is not None or self.M > 1 or getattr(self, "_want_states", False)
        if self._states is None and self._last is not None and want and not self._needs_generic():
            lib = N.lib()
            p = self._last
            if p.noise_src == N.NOISE_KTN:
                self._convert_noise(p)
            K = self.K_local
            states = torch.empty(max(1, self.M), K, self.T, self.nx, device=self.d, dtype=self.dtype)
            scratch = torch.empty(K, device=self.d, dtype=self.dtype)
            old = p.cost_total
            p.states, p.cost_total = _ptr(states), _ptr(scratch)
            N.check(lib.mppi_rollout_cost(C.byref(p), self._stream()), "mppi_rollout_cost")
            p.states, p.cost_total = None, old
            # the rerun rewrote the block minima with identical values; nothing else changed
            self._states = states
        return self._states

    @states.setter
    def states(self, v):
        self._states = v

    @property
    def actions(self):
        want = self.terminal_state_cost is not None or self.M > 1
        if self._actions is None and self._last is not None and want and not self._needs_generic():
            # = (u_scale*v)/u_scale, mppi.py:412; M > 1: the same actions for every rollout copy (:354)
            self._actions = self.perturbed_action.unsqueeze(0).expand(max(1, self.M), -1, -1, -1)
        return self._actions

    @actions.setter
    def actions(self, v):
        self._actions = v

    def _bound_action(self, action):
        return torch.clamp(action, self.u_min, self.u_max)

    def get_rollouts(self, state, num_rollouts=1, U=None):
        """mppi.py:425-448 (off the hot path): (num_rollouts, T, nx) states under the nominal U."""
        state = state.view(-1, self.nx)
        if state.size(0) == 1:
            state = state.expand(num_rollouts, -1)
        if U is None:
            U = self.get_action_sequence()
        T = U.shape[0]
        states = torch.zeros((num_rollouts, T + 1, self.nx), dtype=U.dtype, device=U.device)
        states[:, 0] = state
        for t in range(T):
            next_state = self._dynamics_fn(states[:, t].view(num_rollouts, -1),
                                           self.u_scale * U[t].expand(num_rollouts, -1), t)
            states[:, t + 1] = next_state[:, :self.nx]
        return states[:, 1:]


class GraphedCommand:
    """One captured `command()` (see `MPPI.capture_command`).  `g(state)` copies the state into the
    graph's static input, replays, and returns the graph's static action tensor (overwritten by the
    next replay -- clone it to keep it).  `ctrl.U`, `cost_total` and `omega` refer to the graph's
    static buffers and are current after every replay.  The lazily materialised attributes
    (`noise`, `perturbed_action`, `states`, `actions`) are NOT available under replay (they read
    None): they would have to be re-derived from the nominal sequence the command started from,
    which the replay has already overwritten with the updated one."""

    def __init__(self, ctrl, state, shift, warmup):
        self.ctrl = ctrl
        # rng="torch": torch.randn registers its generator with the graph and replays advance it; the engine's own launch of
        # the same values (MPPI._torch_stream_fill) takes the generator's offset as an argument, which a graph would freeze
        ctrl._in_capture = True
        try:
            self._capture(ctrl, state, shift, warmup)
        finally:
            ctrl._in_capture = False

    def _capture(self, ctrl, state, shift, warmup):
        self.state = ctrl._to_state(state).clone()
        self.U = ctrl.U.detach().to(device=ctrl.d, dtype=ctrl.dtype).clone().contiguous()
        ctrl.U = self.U
        side = torch.cuda.Stream(device=ctrl.d)
        side.wait_stream(torch.cuda.current_stream(ctrl.d))
        with torch.cuda.stream(side):                       # warm-up off the capture: allocator, lazy init
            U_save = self.U.clone()
            for _ in range(max(1, warmup)):
                ctrl.U = self.U
                ctrl.command(self.state, shift_nominal_trajectory=shift)
                self.U.copy_(ctrl.U)
            self.U.copy_(U_save)
        torch.cuda.current_stream(ctrl.d).wait_stream(side)
        torch.cuda.synchronize(ctrl.d)
        self.graph = torch.cuda.CUDAGraph()
        ctrl.U = self.U
        with torch.cuda.graph(self.graph):
            self.action = ctrl.command(self.state, shift_nominal_trajectory=shift)
            self.U.copy_(ctrl.U)                            # hand-over: next replay starts from the new U
        self._U_out = ctrl.U
        ctrl.U = self.U
        torch.cuda.synchronize(ctrl.d)
        self.U.copy_(U_save)                                # the capture pass itself must not advance U

    def __call__(self, state):
        if not torch.is_tensor(state):
            state = torch.tensor(state)
        self.state.copy_(state.to(dtype=self.state.dtype).reshape(self.state.shape), non_blocking=True)
        self.graph.replay()
        c = self.ctrl
        c._last = None                       # see the class docstring: no lazy attributes under replay
        c._noise = c._perturbed_action = c._states = c._actions = None
        c._omega = c._wnz = None             # derived again, on demand, from this replay's cost_total / record
        return self.action


class SMPPI(MPPI):
    """Smooth MPPI (mppi.py:451-570): the sampled quantity U is the action DERIVATIVE, the commanded
    `action_sequence` integrates it, and the cost gains w * |u_scale * diff_t(action)|^2.

    Same kernels as MPPI: the host hands them the base sequence B = action_sequence + U*dt, the
    colouring factors pre-multiplied by dt and the ACTION bounds; the kernels measure the bounded
    noise from B, rescale it by 1/dt ((v - A)/dt - U, :544) and add the smoothness term.
    Reference behaviour kept: the d-action bounds u_min/u_max only shape the stored
    `perturbed_control`, not the rollouts (:536-540).  Deviation: `action_sequence` is re-bound to a
    new tensor per command (the reference updates it in place, :515, which silently rewrites
    actions returned by earlier calls)."""

    def __init__(self, *args, w_action_seq_cost=1., delta_t=1., U_init=None, action_min=None, action_max=None,
                 **kwargs):
        self.w_action_seq_cost = w_action_seq_cost
        self.delta_t = delta_t
        super().__init__(*args, U_init=U_init, **kwargs)
        self.coloured_fill = False     # `perturbed_control` re-derives U + eps from the raw normals (:535-537)
        if action_min is not None and action_max is None:                 # :464-471
            if not torch.is_tensor(action_min):
                action_min = torch.tensor(action_min)
            action_max = -action_min
        if action_max is not None and action_min is None:
            if not torch.is_tensor(action_max):
                action_max = torch.tensor(action_max)
            action_min = -action_max
        if action_min is not None:
            self.action_min = action_min.to(device=self.d)
            self.action_max = action_max.to(device=self.d)
        else:
            self.action_min = torch.tensor(float('-inf'), device=self.d)
            self.action_max = torch.tensor(float('inf'), device=self.d)
        if U_init is None:                                                # :479-483
            self.action_sequence = torch.zeros_like(self.U)
        else:
            self.action_sequence = self.U.clone()
        self.U = torch.zeros_like(self.U)
        self._perturbed_control = None
        self._dt_cache = None

    def get_params(self):
        return f"{super().get_params()} w={self.w_action_seq_cost} t={self.delta_t}"

    def shift_nominal_trajectory(self):
        # roll(-1) + overwrite of the last row (mppi.py:488-492) as ONE concatenation each: these are
        # host-launched tiny kernels and a command is only ~100 us long
        if self.d.type == "cuda" and tuple(self.U.shape) == (self.T, self.nu) == tuple(self.action_sequence.shape):
            # both shifts and the base sequence A + U*dt of the command that follows in ONE small launch
            U = self.U.to(device=self.d, dtype=self.dtype).contiguous()
            A = self.action_sequence.to(device=self.d, dtype=self.dtype).contiguous()
            U_new, A_new, B = torch.empty_like(U), torch.empty_like(U), torch.empty_like(U)
            N.check(N.lib().mppi_smppi_shift(_DT[self.dtype], self.T, self.nu, _ptr(U), _ptr(self._vec(self.u_init)), _ptr(A),
                                             float(self.delta_t), _ptr(U_new), _ptr(A_new), _ptr(B), self._stream()), "mppi_smppi_shift")
            self.U, self.action_sequence = U_new, A_new
            self._base_ready = (U_new, A_new, float(self.delta_t), B)
            return
        u_last = torch.as_tensor(self.u_init, device=self.U.device, dtype=self.U.dtype).reshape(1, -1).expand(1, self.nu)
        self.U = torch.cat((self.U[1:], u_last), dim=0)
        A = self.action_sequence
        self.action_sequence = torch.cat((A[1:], A[-1:]), dim=0)          # :491-492 (last row repeats)

    def get_action_sequence(self):
        return self.action_sequence

    def reset(self):
        self.U = torch.zeros_like(self.U)
        self.action_sequence = torch.zeros_like(self.U)

    def change_horizon(self, horizon):
        if horizon < self.U.shape[0]:
            self.U = self.U[:horizon]
            self.action_sequence = self.action_sequence[:horizon]
        elif horizon > self.U.shape[0]:
            extend_for = horizon - self.U.shape[0]
            self.U = torch.cat((self.U, self.u_init.repeat(extend_for, 1)))
            self.action_sequence = torch.cat((self.action_sequence, self.action_sequence[-1].repeat(extend_for, 1)))
        self.T = horizon
        self._ws = None

    def _bound_d_action(self, control):
        return torch.clamp(control, self.u_min, self.u_max)

    def _bound_action(self, action):
        return torch.clamp(action, self.action_min, self.action_max)

    def _problem(self, Tn=None, U=None):
        p = super()._problem(Tn, U)
        dt = float(self.delta_t)
        keep = p._keep
        A = self.action_sequence.to(device=self.d, dtype=self.dtype)
        br = getattr(self, "_base_ready", None)
        if br is not None and br[0] is self.U and br[1] is self.action_sequence and br[2] == dt and keep["U"] is self.U:
            keep["B"] = br[3]                                             # came out of the shift launch
        else:
            keep["B"] = torch.add(A, keep["U"], alpha=dt).contiguous()    # base of :540, one kernel
        # colouring factors x dt: constant between parameter changes -> cached on the parameter tensors
        ck = (id(keep["L"]), keep["L"]._version, id(keep["mu"]), keep["mu"]._version, dt)
        if self._dt_cache is None or self._dt_cache[0] != ck:
            self._dt_cache = (ck, (keep["L"] * dt).contiguous(), (keep["mu"] * dt).contiguous(), keep["L"], keep["mu"])
        keep["L_dt"], keep["mu_dt"] = self._dt_cache[1], self._dt_cache[2]
        keep["amin"], keep["amax"] = self._vec(self.action_min), self._vec(self.action_max)
        p.base_seq = _ptr(keep["B"])
        p.noise_L, p.noise_mu = _ptr(keep["L_dt"]), _ptr(keep["mu_dt"])
        p.u_min, p.u_max = _ptr(keep["amin"]), _ptr(keep["amax"])
        p.noise_rescale = 1.0 / dt
        p.smooth_weight = float(self.w_action_seq_cost) * float(self.u_scale) ** 2
        return p

    def _prepare(self, state, shift):
        if shift:
            self.shift_nominal_trajectory()       # U and the action sequence move together (host, tiny)
        self._perturbed_control = None
        return super()._prepare(state, False)

    def _end(self, p):
        super()._end(p)
        self.action_sequence = torch.add(self.action_sequence, self.U, alpha=float(self.delta_t))   # :515 (new tensor)
        action = self.action_sequence[:self.u_per_command]
        if self.u_per_command == 1:
            action = action[0]
        return action

    @property
    def perturbed_control(self):
        """clamp(U + eps, u_min, u_max) of the last command (mppi.py:537) -- stored only, unused by
        the rollouts, exactly like the reference."""
        if self._perturbed_control is None and self._last is not None:
            lib = N.lib()
            if self._last.noise_src == N.NOISE_KTN:
                self._convert_noise(self._last)
            q = MPPI._problem(self, U=self._last._keep["U"])          # plain-MPPI view of the same draw
            q.shift = 0
            q.noise_src, q.z, q.call = self._last.noise_src, self._last.z, self._last.call
            q.sample_null_action, q.n_sampler_rows = 0, 0
            self._attach_workspace(q)
            pc = torch.empty(self.K_local, self.T, self.nu, device=self.d, dtype=self.dtype)
            q.perturbed_action = _ptr(pc)
            z_save = None
            if q.noise_src == N.NOISE_PHILOX:
                q.z = None
            N.check(lib.mppi_prepare(C.byref(q), self._stream()), "mppi_prepare")
            self._perturbed_control = pc
        return self._perturbed_control

    @perturbed_control.setter
    def perturbed_control(self, v):
        self._perturbed_control = v


class MPPI_Batched:
    """MPPI for N parallel environments (mppi.py:691-873): N nominal sequences U (N,T,nu), ONE shared
    noise draw per command, independent softmax per environment.

    Here the environment is the z axis of every launch grid: K1/K3/K4 run once for all N
    environments (fused path: native model; generic path: the reference's single (N*K, nx)
    callback batch per timestep around `mppi_prepare`).  Constructor and `command(states)` as in
    the reference; `rng` / `seed` are the same additive extras as on `MPPI`."""

    def __init__(self, dynamics, running_cost, nx, noise_sigma, num_envs,
                 num_samples=100, horizon=15, device="cpu",
                 lambda_=1.,
                 noise_mu=None,
                 u_min=None,
                 u_max=None,
                 u_init=None,
                 u_scale=1,
                 u_per_command=1,
                 step_dependent_dynamics=False,
                 noise_abs_cost=False,
                 *, rng="torch", seed=None, shard=None, auto_jit=None):
        # shard = (rank, world_size[, group]): the ENVIRONMENT axis is split contiguously over the ranks
        # ("the better fit for filling 8 GPUs"); every environment is a complete,
        # independent controller, so a sharded command needs no collective at all -- only the ONE noise
        # draw all environments share (mppi.py:838) must be the same on every rank, which the engine's
        # Philox stream is by construction (a pure function of seed and command number)
        self.N_global = int(num_envs)
        self.env_offset = 0
        self._env_shard = None
        if shard is not None:
            from .dist import ShardPlan
            self._env_shard = ShardPlan(num_envs, *shard)
            if self._env_shard.world_size > 1 and rng not in ("philox", "philox7"):
                raise ValueError("MPPI_Batched(shard=...) needs rng='philox': the shared noise draw must be identical on "
                                 "every rank (or inject it with inject_noise)")
            num_envs = self._env_shard.K_local
            self.env_offset = self._env_shard.k_offset
        # parameter resolution is MPPI's (identical rules, mppi.py:730-790); the inner controller is
        # never commanded itself -- it is the parameter block + launch plumbing for all N envs
        self._c = MPPI(dynamics, running_cost, nx, noise_sigma, num_samples=num_samples, horizon=horizon,
                       device=device, lambda_=lambda_, noise_mu=noise_mu, u_min=u_min, u_max=u_max, u_init=u_init,
                       U_init=torch.zeros(horizon, 1 if len(noise_sigma.shape) == 0 else noise_sigma.shape[0],
                                          dtype=noise_sigma.dtype),
                       u_scale=u_scale, u_per_command=u_per_command, step_dependent_dynamics=step_dependent_dynamics,
                       noise_abs_cost=noise_abs_cost, rng=rng, seed=seed, auto_jit=auto_jit)
        c = self._c
        self.d, self.dtype = c.d, c.dtype
        self.N, self.K, self.T, self.nx, self.nu = num_envs, c.K, c.T, c.nx, c.nu
        self.u_per_command = u_per_command
        self.U = self._initial_U()                                        # :796-797
        self.cost_total = self.omega = None

    def _initial_U(self):
        """(N,T,nu) random nominal sequences (mppi.py:796-797).  Sharded: rank 0's draw for all N_global
        environments, broadcast, of which this rank keeps its slice -- the same U an unsharded
        controller seeded like rank 0 would hold."""
        if self._env_shard is None or self._env_shard.world_size <= 1:
            return self._sample_noise((self.N, self.T))
        c = self._c
        c._shard = self._env_shard                     # borrow MPPI._replicated's broadcast
        try:
            U = c._replicated(self._sample_noise((self.N_global, self.T)))
        finally:
            c._shard = None
        return U[self.env_offset:self.env_offset + self.N].contiguous()

    # attribute surface shared with the inner parameter block
    jit_note = property(lambda self: self._c.jit_note)

    def wait_for_jit(self, timeout=None):
        """see MPPI.wait_for_jit (plain callables traced into fused kernels by a background hipcc run)"""
        return self._c.wait_for_jit(timeout)

    lambda_ = property(lambda self: self._c.lambda_, lambda self, v: setattr(self._c, "lambda_", v))
    u_scale = property(lambda self: self._c.u_scale, lambda self, v: setattr(self._c, "u_scale", v))
    u_min = property(lambda self: self._c.u_min, lambda self, v: setattr(self._c, "u_min", v))
    u_max = property(lambda self: self._c.u_max, lambda self, v: setattr(self._c, "u_max", v))
    u_init = property(lambda self: self._c.u_init, lambda self, v: setattr(self._c, "u_init", v))
    noise_mu = property(lambda self: self._c.noise_mu)
    noise_sigma = property(lambda self: self._c.noise_sigma)
    noise_abs_cost = property(lambda self: self._c.noise_abs_cost)

    def _sample_noise(self, shape):
        return self._c._sample_noise(shape)

    def compile(self, **kwargs):
        self._c.compile(**kwargs)

    def reset(self):
        self.U = self._initial_U()

    def inject_noise(self, z):
        self._c.inject_noise(z)

    def command(self, states, shift_nominal_trajectory=True):
        """states (N,nx) -> actions (N,nu) or (N,u_per_command,nu)   (mppi.py:811-873)"""
        lib = N.lib()
        c = self._c
        if c._jit_pending is not None:
            c._adopt_background_model()
        if not torch.is_tensor(states):
            states = torch.tensor(states)
        states = states.to(dtype=self.dtype, device=self.d)
        if getattr(c._model, "watch", None) is not None:
            c._check_traced(states.reshape(-1, self.nx)[0])
        if self.N != self.N_global and states.numel() == self.N_global * self.nx:
            states = states.reshape(self.N_global, self.nx)[self.env_offset:self.env_offset + self.N]   # this rank's environments
        states = states.reshape(self.N, self.nx).contiguous()
        Nn, K, T, nu = self.N, self.K, self.T, self.nu
        p = c._problem(U=self.U.reshape(Nn * T, nu))
        p.num_envs = Nn
        p.shift = int(bool(shift_nominal_trajectory))
        st = c._stream()
        c._attach_workspace(p)
        c._draw_noise(p, (K, T, nu))                                      # shared across environments (:838)
        if p.noise_src == N.NOISE_PHILOX:
            # ONE draw serves all N environments: generate the rows once, every environment's K1 / K3
            # block then reads them (in-kernel generation would repeat the Philox work N times)
            if not p.z:
                zn = torch.empty(c._zelems(T), device=self.d, dtype=self.dtype)
                p.z = _ptr(zn)
                p._keep["z"] = zn
            N.check(lib.mppi_noise_fill_philox(C.byref(p), p.z, st), "mppi_noise_fill_philox")
            p.noise_src = N.NOISE_TNK4
        cost_total = torch.empty(Nn, K, device=self.d, dtype=self.dtype)
        p.cost_total = _ptr(cost_total)
        p.state = _ptr(states)
        p._keep["state"] = states
        if not c._needs_generic():
            N.check(lib.mppi_rollout_cost(C.byref(p), st), "mppi_rollout_cost")
        else:
            pa = torch.empty(Nn, K, T, nu, device=self.d, dtype=self.dtype)
            pert = torch.empty(Nn, K, device=self.d, dtype=self.dtype)
            p.perturbed_action, p.pert_cost = _ptr(pa), _ptr(pert)
            N.check(lib.mppi_prepare(C.byref(p), st), "mppi_prepare")
            p.perturbed_action = p.pert_cost = None
            NK = Nn * K
            state = states.unsqueeze(1).expand(Nn, K, self.nx).reshape(NK, self.nx)   # :848-850
            rollout = torch.zeros(Nn, K, device=self.d, dtype=self.dtype)
            with torch.no_grad():                                         # (see _generic_total_cost)
                for t in range(T):
                    u = c.u_scale * pa[:, :, t].reshape(NK, nu)
                    state = c._dynamics_fn(state, u, t)
                    rollout = rollout + c._running_cost_fn(state, u, t).reshape(Nn, K)
            torch.add(rollout, pert, out=cost_total)                      # :861
            N.check(lib.mppi_cost_block_min(C.byref(p), st), "mppi_cost_block_min")
        if p.noise_src == N.NOISE_PHILOX and p.z:
            p.noise_src = N.NOISE_TNK4
        omega = torch.empty(Nn, K, device=self.d, dtype=self.dtype)
        U_new = torch.empty(Nn, T, nu, device=self.d, dtype=self.dtype)
        record = torch.empty(Nn, 2 + T * nu, device=self.d, dtype=self.dtype)
        p.omega, p.U_out, p.record = _ptr(omega), _ptr(U_new), _ptr(record)
        N.check(lib.mppi_weights_partial(C.byref(p), st), "mppi_weights_partial")   # per-env beta/eta (:863-866)
        N.check(lib.mppi_finalize(C.byref(p), 1, st), "mppi_finalize")
        self.cost_total, self.omega, self._last = cost_total, omega, p
        self.U = U_new                                                    # :869
        action = self.U[:, :self.u_per_command]
        if self.u_per_command == 1:
            action = action[:, 0]
        return action


class TimeKernel:
    """mppi.py:573-577"""

    def __call__(self, t, tk):
        raise NotImplementedError


class RBFKernel(TimeKernel):
    """mppi.py:580-590"""

    def __init__(self, sigma=1):
        self.sigma = sigma

    def __repr__(self):
        return f"RBFKernel(sigma={self.sigma})"

    def __call__(self, t, tk):
        d = torch.sum((t[:, None] - tk) ** 2, dim=-1)
        return torch.exp(-d / (1e-8 + 2 * self.sigma ** 2))


class KMPPI(MPPI):
    """MPPI with kernel interpolation of control points (mppi.py:593-688).

    The reference solves K identical (S,S) systems under vmap each command; every sample sees
    the same `Tk`/`Hs`, so the interpolation is one constant operator W = K(Hs,Tk) Ktktk^-1
    (T,S) -- built once on the host here, applied in `mppi_kmppi_interp`."""

    def __init__(self, *args, num_support_pts=None, kernel: TimeKernel = RBFKernel(), **kwargs):
        super().__init__(*args, **kwargs)
        self.num_support_pts = num_support_pts or self.T // 2
        self.theta = torch.zeros((self.num_support_pts, self.nu), dtype=self.dtype, device=self.d)
        self.interpolation_kernel = kernel
        self.ktn_direct = False        # the support-point draw always goes through the layout conversion
        self.coloured_fill = False     # the interpolation kernel colours the support points itself
        self.fuse_interpolation = True  # K1 interpolates in-kernel where it can (mppi_rollout_cost_kmppi)
        self.onchip_update = True       # ... and reduces its part of the theta update from the control points it holds (mppi_command_kmppi)
        self._noise_theta = None
        self._last_theta = None
        self.prepare_vmap_interpolation()

    def get_params(self):
        return f"{super().get_params()} num_support_pts={self.num_support_pts} kernel={self.interpolation_kernel}"

    def reset(self):
        super().reset()
        self.theta.zero_()

    def change_horizon(self, horizon):
        """The reference inherits MPPI.change_horizon and leaves Tk/Hs stale (next command raises
        a shape error); here the operators are rebuilt."""
        super().change_horizon(horizon)
        self.prepare_vmap_interpolation()

    def prepare_vmap_interpolation(self):
        """Name kept from mppi.py:636-651; builds Tk, Hs and the constant operators."""
        S = int(self.num_support_pts)
        tk = torch.linspace(0, self.T - 1, S, device=self.d, dtype=self.dtype)
        hs = torch.linspace(0, self.T - 1, int(self.T), device=self.d, dtype=self.dtype)
        self.Tk = tk.unsqueeze(0).repeat(self.K, 1)
        self.Hs = hs.unsqueeze(0).repeat(self.K, 1)
        k = self.interpolation_kernel
        Ktktk = k(tk.unsqueeze(-1), tk.unsqueeze(-1))
        self._W = torch.linalg.solve(Ktktk, k(hs.unsqueeze(-1), tk.unsqueeze(-1)), left=False).contiguous()
        self._W_shift = torch.linalg.solve(Ktktk, k((tk + 1).unsqueeze(-1), tk.unsqueeze(-1)), left=False).contiguous()

    def do_kernel_interpolation(self, t, tk, c):
        K = self.interpolation_kernel(t.unsqueeze(-1), tk.unsqueeze(-1))
        Ktktk = self.interpolation_kernel(tk.unsqueeze(-1), tk.unsqueeze(-1))
        KK = torch.linalg.solve(Ktktk, K, left=False)
        return torch.matmul(KK, c), K

    def deparameterize_to_trajectory_single(self, theta):
        return self.do_kernel_interpolation(self.Hs[0], self.Tk[0], theta)

    def deparameterize_to_trajectory_batch(self, theta):
        assert theta.shape == (self.K, self.num_support_pts, self.nu)
        K = self.interpolation_kernel(self.Hs[0].unsqueeze(-1), self.Tk[0].unsqueeze(-1))
        return torch.einsum("ts,ksn->ktn", self._W, theta), K.unsqueeze(0).expand(self.K, -1, -1)

    def shift_nominal_trajectory(self):
        if not self._native_sequences():
            super().shift_nominal_trajectory()
            self.theta = self._W_shift @ self.theta                       # mppi.py:617-619
            return
        # one small launch for both sequences (host-side: roll + copy + GEMM = three)
        U = self.U.to(device=self.d, dtype=self.dtype).contiguous()
        th = self.theta.to(device=self.d, dtype=self.dtype).contiguous()
        u0 = self._vec(self.u_init)
        U_new, th_new = torch.empty_like(U), torch.empty_like(th)
        N.check(N.lib().mppi_kmppi_shift(_DT[self.dtype], self.T, int(self.num_support_pts), self.nu, _ptr(U), _ptr(u0), _ptr(th),
                                         _ptr(self._W_shift), _ptr(U_new), _ptr(th_new), self._stream()), "mppi_kmppi_shift")
        self.U, self.theta = U_new, th_new

    def _fused_interp_expected(self):
        """mirror of mppi_rollout_cost_kmppi's conditions (include/mppi_amd.h); a wrong guess only costs time"""
        S, nu = int(self.num_support_pts), self.nu
        return (self.fuse_interpolation and self.dtype == torch.float32 and self._diagonal_sigma and nu % 4 == 0
                and nu <= 16 and S <= min(64, (384 // nu) & ~3) and not self._needs_generic()
                and not isinstance(self._model, MLPResidual))

    def _native_sequences(self):
        return (self.d.type == "cuda" and self._W.dtype == self.dtype and tuple(self.U.shape) == (self.T, self.nu)
                and tuple(self.theta.shape) == (int(self.num_support_pts), self.nu))

    def _trajectory_of(self, theta):
        """U = W theta (mppi.py:682)"""
        if not self._native_sequences():
            return self._W @ theta
        U = torch.empty(self.T, self.nu, device=self.d, dtype=self.dtype)
        N.check(N.lib().mppi_kmppi_trajectory(_DT[self.dtype], self.T, int(self.num_support_pts), self.nu, _ptr(self._W),
                                              _ptr(theta), _ptr(U), self._stream()), "mppi_kmppi_trajectory")
        return U

    def _noise_shape(self):
        return (self.K_local, int(self.num_support_pts), self.nu)

    def _prepare(self, state, shift):
        """the host part of a KMPPI command (MPPI._prepare): support-point draw, the trajectory problem `p` and the THETA problem
        `pt` (K3 / K4 run on the support-point stream, mppi.py:679-681); returns `pt` -- its record is what a sharded command
        exchanges -- with `pt._traj = p`"""
        self.state = self._to_state(state)
        if shift:
            # explicit shift (tiny (T,nu)/(S,S) host-launched ops) so that theta and U move together
            self.shift_nominal_trajectory()
        S = int(self.num_support_pts)
        K = self.K_local
        st = self._stream()
        # --- support-point stream problem: "sequence" = theta (S,nu) ---
        pt = self._problem(Tn=S, U=self.theta)
        pt.shift = 0
        pt.sample_null_action = 0
        self._attach_workspace(pt)
        fill_keep = self.philox_fill
        if self.philox_fill is None and self._fused_interp_expected() and self.philox_rounds != 7:
            # K1 keeps the bounded control points in registers: generating their rows there (and again in K3)
            # costs +4 us of K1 and saves the 19 us generator launch and its 100 MB (C3-sized work)
            self.philox_fill = False
        try:
            self._draw_noise(pt, self._noise_shape())
        finally:
            self.philox_fill = fill_keep
        if pt.noise_src == N.NOISE_PHILOX:
            pt.z = None       # support-point stream is tiny: interp and the theta update regenerate it
        # --- trajectory problem ---
        p = self._problem()
        p.shift = 0
        p.S = S
        p.theta = pt.U
        p._keep["theta_keep"] = pt._keep
        p.W = _ptr(self._W)
        p.noise_src, p.z, p.call = pt.noise_src, pt.z, pt.call
        self._attach_workspace(p)
        pt.workspace, pt.workspace_elems = p.workspace, p.workspace_elems
        self._sampler_rows(p)
        cost_total = torch.empty(K, device=self.d, dtype=self.dtype)
        p.cost_total = _ptr(cost_total)
        per_sample = tuple(self.state.shape) == (K, self.nx)
        self._states = self._actions = self._noise = self._perturbed_action = None
        self._noise_theta = None
        sharded = self._sharded()
        # omega = (1/eta) exp(-(c - beta)/lambda) and cost_total_non_zero are functions of cost_total and the record: a
        # single-shard command leaves them to their first read (MPPI.omega); a sharded one has K5 rescale them
        lazy = not sharded
        omega = None if lazy else torch.empty(K, device=self.d, dtype=self.dtype)
        wnz = None if lazy else torch.empty(K, device=self.d, dtype=self.dtype)
        theta_new = torch.empty(S, self.nu, device=self.d, dtype=self.dtype)
        record = torch.empty(2 + S * self.nu, device=self.d, dtype=self.dtype)
        pt.cost_total = p.cost_total
        pt.omega, pt.cost_total_non_zero, pt.U_out, pt.record = _ptr(omega), _ptr(wnz), _ptr(theta_new), _ptr(record)
        pt.u_per_command = 0
        self.cost_total = cost_total
        # the record of the exchange (MPPI._command / group.DeviceGroup) is the THETA problem's: {beta, eta, P_theta[S nu]}
        pt._keep.update(record=record, omega=omega, wnz=wnz, theta_new=theta_new, lazy=lazy)
        pt._traj = p        # (an attribute of the block, NOT an entry of pt._keep: p._keep["theta_keep"] IS that dictionary, and a reference
        #                      cycle would keep every command's buffers -- 200 MB of raw actions in the two-launch form -- alive until
        #                      the cycle collector runs: fresh hipMallocs per command in the meantime, 0.7 ms each)
        pt._apply = 0 if sharded else 1
        if not self._needs_generic():
            s0 = self._fused_state(per_sample)
            p.state = _ptr(s0)
            p._keep["state"] = s0
            p.state_per_sample = int(per_sample)
            p.use_terminal = int(self.terminal_state_cost is not None)
            pt._deferred = True
            return pt
        pt._deferred = False
        self._raw_actions(p)
        self._generic_total_cost(p, cost_total, st)
        self._theta_update(pt, st)
        return pt

    def _theta_update(self, pt, st):
        """K3 / K4 on the support-point stream (mppi.py:679-681), stand-alone"""
        lib = N.lib()
        N.check(lib.mppi_weights_partial(C.byref(pt), st), "mppi_weights_partial")
        N.check(lib.mppi_finalize(C.byref(pt), pt._apply, st), "mppi_finalize")
        self._settle_next()

    def _group_blocks(self, pt):
        """what a device group's worker issues for this prepared command (csrc/group.hip: mppi_command_kmppi(trajectory problem,
        theta problem)) -- or None: this command has no one-call form, the shard launches it itself"""
        if self.fuse_interpolation and self.onchip_update:
            return pt._traj, pt
        return None

    def _launch_prepared(self, pt):
        """the fused path's launches of a prepared KMPPI command, on the calling thread"""
        lib, st, p = N.lib(), self._stream(), pt._traj
        # interpolation inside K1 where that kernel exists (fp32, diagonal Sigma, nu % 4 == 0, S*nu <= 384):
        # the (K,T,nu) raw actions are never written; lazy attributes build them on demand (_raw_actions).
        # ONE call for the command (mppi_command_kmppi): where it can, that kernel also reduces its workgroups' part of
        # the theta update from the control points the lanes still hold, and the stand-alone K3 -- which re-creates all
        # S*nu control-point rows per sample -- is replaced by the small combine launch of the on-chip MPPI command
        updated = False
        if not self.fuse_interpolation:
            rc = N.E_UNSUPPORTED
        elif self.onchip_update:
            rc = lib.mppi_command_kmppi(C.byref(p), C.byref(pt), pt._apply, st)
            updated = rc == 0
        else:
            rc = lib.mppi_rollout_cost_kmppi(C.byref(p), st)      # (A/B seam: K1 here, the stand-alone K3 / K4 below)
        if rc == N.E_UNSUPPORTED:
            self._raw_actions(p)
            N.check(lib.mppi_rollout_cost(C.byref(p), st), "mppi_rollout_cost")
        else:
            N.check(rc, "mppi_command_kmppi")
        if updated:
            self._launched(pt, 0, int(lib.mppi_last_next_draw()))
        else:
            pt._deferred = False
            self._theta_update(pt, st)

    def _launched(self, pt, form, next_draw):
        pt._deferred = False
        self._settle_next(next_draw)

    def _end(self, pt):
        p = pt._traj
        self._omega, self._wnz = pt._keep["omega"], pt._keep["wnz"]
        record = pt._keep["record"]
        self._lazy_w = (float(self.lambda_), record) if pt._keep["lazy"] else None
        self._record = record
        self._last, self._last_theta = p, pt
        self.theta = pt._keep["theta_new"]
        self.U = self._trajectory_of(self.theta)                          # mppi.py:682
        action = self.U[:self.u_per_command]
        if self.u_per_command == 1:
            action = action[0]
        return action

    def _raw_actions(self, p=None):
        """(K,T,nu) raw interpolated actions of the last command in the engine layout (mppi.py:665): the
        two-launch form of K1, the generic path and the lazy attributes read them; the fused K1 does not."""
        p = self._last if p is None else p
        if p is None or "v_raw" in p._keep:
            return
        v_raw = torch.empty(self._zelems(self.T), device=self.d, dtype=self.dtype)
        N.check(N.lib().mppi_kmppi_interp(C.byref(p), _ptr(v_raw), self._stream()), "mppi_kmppi_interp")
        p.noise_src, p.z = N.NOISE_ACTIONS, _ptr(v_raw)
        p._keep["v_raw"] = v_raw

    def _materialize(self):
        self._raw_actions()
        super()._materialize()

    @property
    def states(self):
        self._raw_actions()
        return MPPI.states.fget(self)

    @states.setter
    def states(self, v):
        self._states = v

    @property
    def noise_theta(self):
        """(K,S,nu) bounded control-point noise (mppi.py:664), materialised on first read."""
        if self._noise_theta is None and self._last_theta is not None:
            lib = N.lib()
            pt = self._last_theta
            K, S, nu = self.K_local, int(self.num_support_pts), self.nu
            nt = torch.empty(K, S, nu, device=self.d, dtype=self.dtype)
            pt.noise = _ptr(nt)
            N.check(lib.mppi_prepare(C.byref(pt), self._stream()), "mppi_prepare")
            pt.noise = None
            self._noise_theta = nt
        return self._noise_theta

    @noise_theta.setter
    def noise_theta(self, v):
        self._noise_theta = v


def run_mppi(mppi, env, retrain_dynamics, retrain_after_iter=50, iter=1000, render=True):
    """Closed-loop helper with the reference's contract (mppi.py:876-898): step a gym-style `env`
    `iter` times with `mppi.command(env.unwrapped.state)`, keep the last `retrain_after_iter`
    (state, action) rows in a device tensor, hand that tensor to `retrain_dynamics` every
    `retrain_after_iter` steps, return (total reward, dataset).  Host glue around `command()`;
    the only device->host transfer per step is the action the environment needs."""
    import time
    rows = retrain_after_iter
    dataset = torch.zeros((rows, mppi.nx + mppi.nu), dtype=mppi.U.dtype, device=mppi.d)
    total_reward = 0
    for i in range(iter):
        state = env.unwrapped.state.copy()
        t0 = time.perf_counter()
        action = mppi.command(state)
        dt = time.perf_counter() - t0
        step_result = env.step(action.cpu().numpy())
        reward = step_result[1]
        total_reward += reward
        logger.debug("step %d: reward %.4f, command() %.5fs", i, float(reward), dt)
        if render:
            env.render()
        row = i % rows
        if row == 0 and i > 0:
            retrain_dynamics(dataset)
            dataset.zero_()
        dataset[row, :mppi.nx] = torch.as_tensor(state, dtype=mppi.U.dtype)
        dataset[row, mppi.nx:] = action
    return total_reward, dataset
