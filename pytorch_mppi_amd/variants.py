"""The rest of the reference's controller family on the same kernels: `SMPPI`
(/root/reference/src/pytorch_mppi/mppi.py:451-570),
`KMPPI` with its `TimeKernel` / `RBFKernel` (:573-688), `MPPI_Batched` (:691-873) and the closed-loop helper `run_mppi`
(:876-898)."""
import ctypes as C
import logging
import os

import torch

from . import _native as N
from ._util import _DT, _ptr
from .controller import MPPI
from .models import MLPResidual

logger = logging.getLogger(__name__)


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
            N.check(N.lib().mppi_smppi_shift(_DT[self.dtype], self.T, self.nu, _ptr(U), _ptr(self._vec(self.u_init)),
                    _ptr(A),
                                             float(self.delta_t), _ptr(U_new), _ptr(A_new), _ptr(B), self._stream()),
                                                     "mppi_smppi_shift")
            self.U, self.action_sequence = U_new, A_new
            self._base_ready = (U_new, A_new, float(self.delta_t), B)
            return
        u_last = torch.as_tensor(self.u_init, device=self.U.device, dtype=self.U.dtype).reshape(1, -1).expand(1,
                self.nu)
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
                raise ValueError("MPPI_Batched(shard=...) needs rng='philox': the shared noise draw must be identical "
                        "on "
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
            # this rank's environments
            states = states.reshape(self.N_global, self.nx)[self.env_offset:self.env_offset + self.N]
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
        # ... and reduces its part of the theta update from the control points it holds (mppi_command_kmppi)
        self.onchip_update = True
        # the next command's shifted sequences beside U = W theta
        self.shift_ahead = os.environ.get("MPPI_KMPPI_SHIFT_AHEAD", "1") != "0"
        self._shift_ready = None
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
        sr, self._shift_ready = self._shift_ready, None
        if (sr is not None and sr[0] is self.U and sr[1] is self.theta and sr[2] == self._shift_key()
                and sr[5] == (self.U._version, self.theta._version)):
            # the previous command's last launch made these beside U = W theta (mppi_kmppi_after_update): nothing to
            # launch
            self.U, self.theta = sr[3], sr[4]
            return
        # one small launch for both sequences (host-side: roll + copy + GEMM = three)
        U = self.U.to(device=self.d, dtype=self.dtype).contiguous()
        th = self.theta.to(device=self.d, dtype=self.dtype).contiguous()
        u0 = self._vec(self.u_init)
        U_new, th_new = torch.empty_like(U), torch.empty_like(th)
        N.check(N.lib().mppi_kmppi_shift(_DT[self.dtype], self.T, int(self.num_support_pts), self.nu, _ptr(U),
                _ptr(u0), _ptr(th),
                                         _ptr(self._W_shift), _ptr(U_new), _ptr(th_new), self._stream()),
                                                 "mppi_kmppi_shift")
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

    def _shift_key(self):
        """what the shifted sequences depend on besides U and theta themselves"""
        u0 = self.u_init
        return (id(self._W_shift), id(u0), u0._version if torch.is_tensor(u0) else u0, self.T,
                int(self.num_support_pts))

    def _trajectory_of(self, theta):
        """U = W theta (mppi.py:682)"""
        if not self._native_sequences():
            return self._W @ theta
        if self.shift_ahead and tuple(theta.shape) == (int(self.num_support_pts), self.nu):
            # ... and, in the same launch, both sequences as the next command's shift wants them
            S = int(self.num_support_pts)
            U = torch.empty(self.T, self.nu, device=self.d, dtype=self.dtype)
            U_s, th_s = torch.empty_like(U), torch.empty(S, self.nu, device=self.d, dtype=self.dtype)
            N.check(N.lib().mppi_kmppi_after_update(_DT[self.dtype], self.T, S, self.nu, _ptr(self._W),
                    _ptr(self._W_shift), _ptr(theta),
                                                    _ptr(self._vec(self.u_init)), _ptr(U), _ptr(th_s), _ptr(U_s),
                                                            self._stream()),
                    "mppi_kmppi_after_update")
            # (valid while U and theta are these very tensors, unwritten: reset() zeroes theta in place)
            self._shift_ready = (U, theta, self._shift_key(), U_s, th_s, (U._version, theta._version))
            return U
        U = torch.empty(self.T, self.nu, device=self.d, dtype=self.dtype)
        N.check(N.lib().mppi_kmppi_trajectory(_DT[self.dtype], self.T, int(self.num_support_pts), self.nu,
                _ptr(self._W),
                                              _ptr(theta), _ptr(U), self._stream()), "mppi_kmppi_trajectory")
        return U

    def _noise_shape(self):
        return (self.K_local, int(self.num_support_pts), self.nu)

    def _prepare(self, state, shift):
        """the host part of a KMPPI command (MPPI._prepare): support-point draw, the trajectory problem `p` and the
        THETA problem
        `pt` (K3 / K4 run on the support-point stream, mppi.py:679-681); returns `pt` -- its record is what a sharded
        command
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
        # the record of the exchange (MPPI._command / group.DeviceGroup) is the THETA problem's: {beta, eta, P_theta[S
        # nu]}
        pt._keep.update(record=record, omega=omega, wnz=wnz, theta_new=theta_new, lazy=lazy)
        # (an attribute of the block, NOT an entry of pt._keep: p._keep["theta_keep"] IS that dictionary, and a
        # reference
        pt._traj = p
        #                      cycle would keep every command's buffers -- 200 MB of raw actions in the two-launch form
        #                      -- alive until
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
        """what a device group's worker issues for this prepared command (csrc/group.hip: mppi_command_kmppi(trajectory
        problem,
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
