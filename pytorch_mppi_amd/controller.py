"""`MPPI` with the reference's constructor and `.command(state)` surface
(/root/reference/src/pytorch_mppi/mppi.py:35-448), backed by the HIP engine.

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

The class is assembled from four modules: this one (API, state, one command), `draws.py` (noise modes, row buffers,
draw-ahead), `forms.py` (problem block, parameter vectors, fused-or-callback decision, the callback rollout) and
`jit_glue.py`
(traced callables); `variants.py` holds SMPPI / KMPPI / MPPI_Batched.
"""
import ctypes as C
import os
import typing

import torch

from . import _native as N
from ._util import _DT, SpecificActionSampler, _ptr
from .draws import Draws
from .forms import Forms
from .jit_glue import JitGlue, _auto_jit_mode
from .models import native_model_of


class MPPI(Draws, Forms, JitGlue):
    """Model Predictive Path Integral control (Williams et al. 2017, alg. 2), drop-in for
    `pytorch_mppi.MPPI` on MI355X."""

    def __new__(cls, *args, devices=None, **kw):
        # devices=[d0, d1, ...] (two or more): ONE Python process commanding on several GPUs -- the object is a device
        # group (pytorch_mppi_amd/group.py: one shard controller per device, a subclass of `cls`); SURVEY.md 8b / 8e
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
        # rng="philox7": the engine's generator with Philox4x32-7 (Random123's philox4x32_R<7>: the fewest rounds that
        # pass BigCrush) instead of -10 -- another stream, everything else as rng="philox"; 30 % fewer of the multiplies
        # the on-chip command's time is made of (MppiProblem.philox_rounds; oracle/philox.py `rounds`)
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
        # how the last command got its normals: "philox-fill" | "philox-k1" | None (other modes)
        self.last_draw = None
        # rng="philox", full Sigma: let the generator launch apply chol(Sigma) (see _draw_noise)
        self.coloured_fill = True
        # rng="philox": generate in a separate launch (True) / inside K1 (False) / by horizon (None)
        self.philox_fill = None
        # rng="philox": the on-chip command (csrc/rollout_onchip.hpp) -- no (K,T,nu) array at all: one launch generates,
        # rolls out, keeps the bounded noise in accumulation registers / LDS and leaves one partial record per
        # workgroup, a second one combines them.  None: whenever the problem is in its scope (fp32, diagonal Sigma,
        # plain MPPI, M = 1, no sampler rows) and too large for the single-launch form; True / False: force / forbid.
        # (a full Sigma is coloured in the lane: L z + mu per timestep out of LDS)
        self.philox_onchip = None
        self._onchip_refused = False
        # rng="torch": read (K,T,nu) in place when possible
        self.ktn_direct = os.environ.get("MPPI_KTN_DIRECT", "1") != "0"
        # the on-chip command: let what fits neither registers nor LDS wait in memory (one array per controller,
        # allocated on first use) instead of generating it twice
        self.onchip_spill = os.environ.get("MPPI_ONCHIP_SPILL", "1") != "0"
        self._spill = None
        # rng="torch": compute torch.randn's values straight into the engine's rows (see _torch_stream_fill); off: call
        # torch.randn and read / convert its (K,T,nu) array
        self.torch_rows = os.environ.get("MPPI_TORCH_ROWS", "1") != "0"
        # ... and the NEXT command's draw inside this command's K3 launch (ABI 21; adopted at the next command when the
        # generator is where that assumed: _torch_stream_fill).  Costs a second row buffer
        self.draw_ahead = os.environ.get("MPPI_DRAW_AHEAD", "1") != "0"
        # (draws of fewer normals than this keep their own tiny launch: carving them into K3's few workgroups costs more
        # than it saves
        # -- profiles/r05_small_k_sweep.txt)
        self.draw_ahead_min = int(os.environ.get("MPPI_DRAW_AHEAD_MIN", str(1 << 19)))
        # the same for the ENGINE's generator (rng="philox" with rows in memory, MPPI_NEXT_DRAW_PHILOX): built,
        # bit-exact, and OFF -- that generator launch is already bound by its 201 MB of stores (34 us at C3), not by the
        # VALU, and a launch that reads K3's rows while it writes the next ones moves the same 403 MB slower (mixed
        # traffic: 5.3 TB/s against 5.9 one after the other;
        # C3 rows-in-memory command 0.1186 ms with, 0.1096 without; profiles/r05_draw_ahead_forms.txt)
        self.draw_ahead_philox = os.environ.get("MPPI_DRAW_AHEAD_PHILOX", "0") == "1"
        # (shape key, generator, seed, offset, rows): generated, waiting for the next command
        self._next_draw = None
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
        self._traced_user_callables = ((m._dyn, m._cost, m._term) if self._model is not None and getattr(m, "_code",
                None) is not None
                                       and getattr(m, "watch", None) is not None else None)
        self.jit_note = None
        self._jit_pending = None
        # auto_jit: True / "sync" = trace and compile now (construction blocks for the hipcc run unless the object is
        # cached); "async" = trace now, compile in a background thread -- commands run the callbacks until the fused
        # kernels are there; False / "0" = off.  None: the environment's MPPI_AUTO_JIT (default "async")
        mode = _auto_jit_mode(auto_jit if auto_jit is not None else os.environ.get("MPPI_AUTO_JIT", "async"))
        # traced callables are re-checked against the live ones: a flat watch of the places they can read from on every
        # command (watch.StateWatch), and functor-against-callables on a small random batch on the device at adoption
        # and every MPPI_JIT_CHECK_EVERY commands (default 256; 0 = never) -- _check_traced / _spot_check below
        self._jit_mode = mode
        self._jit_check_every = int(os.environ.get("MPPI_JIT_CHECK_EVERY", "256"))
        self._jit_cmds = 0             # fused commands since the current traced model was adopted
        # share of the issuing time the spot-checks may take: the interval is stretched beyond `_jit_check_every` where
        # a check (~1 ms) would cost more than this (0: never stretched)
        self._jit_check_share = float(os.environ.get("MPPI_JIT_CHECK_SHARE", "0.01"))
        self._jit_next_check = 0       # ... and the command at which the next on-device spot-check is due
        self._jit_last_check = None    # (command number, time) of the previous one
        self._jit_retraces = 0         # times the callables' state moved in a way that changed the functor
        self._jit_benign = 0           # ... in a way that did not
        self._jit_spot_checks = 0
        # places (watch.Path) whose tensors were seen to change: run-time parameters from then on
        self._jit_dynamic = []
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
        s = (f"K={self.K} T={self.T} M={self.M} lambda={self.lambda_} noise_mu={self.noise_mu.cpu().numpy()} "
             f"noise_sigma={self.noise_sigma.cpu().numpy()}")
        return s.replace("\n", ",")

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
            # (a device group's shards live in ONE process: group.py copies shard 0's sequences)
            return t
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
    # command
    # ------------------------------------------------------------------------------------------
    def command(self, state, shift_nominal_trajectory=True, info=None):
        """mppi.py:240-252: returns the (nu,) / (u_per_command,nu) action as a device tensor,
        without synchronising."""
        self.info = info
        if self._jit_pending is not None:
            # only HERE, between two commands: a command never changes path half-way
            self._adopt_background_model()
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

    def _command(self, state, shift):
        p = self._begin(state, shift)
        if self._sharded() and not getattr(p, "_combined", False):
            if getattr(self._shard, "local", False):
                raise RuntimeError("this controller is one shard of a device group (MPPI(..., devices=[...])): "
                        "command the group, not the shard")
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

    def _sharded(self):
        # _force_collective: measurement seam (tools/shard_overhead.py) -- run record -> all_gather -> K5 at world_size
        # 1
        return self._shard is not None and (self._shard.world_size > 1 or self._force_collective)

    def _begin(self, state, shift):
        """Everything local to this shard: noise, K1 (or the generic callback loop), K3, K4.
        Single shard: K4 also applies the update.  Sharded: K4 only writes the shard record."""
        p = self._prepare(state, shift)
        if p._deferred:
            self._launch_prepared(p)
        return p

    def _prepare(self, state, shift):
        """The host part of a command up to (not including) the fused path's launch: the problem block with this
        command's draw, buffers and state bound.  `p._deferred`: the fused launch is still to be issued -- by
        `_launch_prepared` on
        this thread, or by the device group's worker thread of this shard's device (group.py, csrc/group.hip).  The
        generic (callback) path cannot be handed over: it has run K1's stand-in, K3 and K4 when this returns
        (`_deferred`
        False)."""
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
        """the fused path's launches of a prepared command, on the calling thread: K1 + K3 + K4 from one C call (a
        sharded rank
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
        """what a device group's worker issues for this prepared command (csrc/group.hip): (problem, theta problem |
        None)"""
        return p, None

    def _launched(self, p, form, next_draw):
        """behind the fused launches of a command (issued here or by the device group's worker): what the engine
        reported"""
        p._deferred = False
        self._settle_next(next_draw)
        if self.last_draw == "philox-onchip" and form != N.FORM_ONCHIP:
            # the engine ran K1 + K3 with the rows generated twice instead (a model without the on-chip kernel, ...):
            # correct, slower -- store the rows from the next command on
            self._onchip_refused = True
            self.last_draw = "philox-twice"
        if p.noise_src == N.NOISE_PHILOX and p.z:
            p.noise_src = N.NOISE_TNK4        # the rows K1 generated are in p.z now (lazy attributes)

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
        # rng="torch": torch.randn registers its generator with the graph and replays advance it; the engine's own
        # launch of the same values (MPPI._torch_stream_fill) takes the generator's offset as an argument, which a graph
        # would freeze
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
