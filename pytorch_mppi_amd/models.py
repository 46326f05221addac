"""Native model specs: the objects that let `MPPI` fuse the rollout.

The reference's plugin API is "any callable": ``dynamics(state, u[, t])``,
``running_cost(state, u[, t])``, ``terminal_state_cost(states, actions)``
(/root/reference/src/pytorch_mppi/mppi.py:63-64, :314, :318, :325).  A Python callable cannot be
compiled into a HIP kernel, so each native model is ONE object that is both

* ordinary torch callables in exactly that convention -- ``model.dynamics``,
  ``model.running_cost``, ``model.terminal_state_cost`` -- usable with the reference `MPPI`
  or with this engine's generic (callback) path, and
* a ``model_id`` + parameter blob for the device functor of the same formula in
  ``csrc/models.hpp``.

``MPPI(model.dynamics, model.running_cost, ...)`` detects the bound methods and launches the
fused kernel; any other callable takes the generic path (kernels around a Python T-loop).
"""
import math

import torch

from . import _native as N


class NativeModel:
    """The built-in models are time-invariant: their callables accept and ignore the timestep argument of
    `step_dependent_dynamics=True` controllers (mppi.py:147-154), so the same fused kernel serves both settings
    (`step_dependent = None`: either)."""
    model_id = N.MODEL_NONE
    step_dependent = None
    nx = 0
    nu = 0
    hidden = 0
    has_terminal = False

    def __init__(self):
        self._blob_cache = {}
        self._param_version = 0
        self.process_noise = None      # (nx,) std of the Gaussian disturbance added to every post-dynamics state

    def with_process_noise(self, std):
        """Make the dynamics stochastic: x' = f(x, u) + std * n, n ~ N(0, I) -- what `rollout_samples > 1`
        (mppi.py:334-373) is for.  The torch callable draws n with torch.randn_like; the fused kernel draws
        it from the engine's Philox stream (one independent draw per sample, timestep and rollout copy)."""
        std = torch.as_tensor(std, dtype=torch.float64).reshape(-1)
        if std.numel() == 1:
            std = std.expand(self.nx).clone()
        if std.numel() != self.nx:
            raise ValueError(f"process noise std must have nx = {self.nx} entries")
        self.process_noise = std
        self._param_version += 1
        return self

    def _noisy(self, nxt):
        if self.process_noise is None:
            return nxt
        return nxt + self.process_noise.to(device=nxt.device, dtype=nxt.dtype) * torch.randn_like(nxt)

    # -- torch callables in the reference's plugin convention ----------------------------------
    def dynamics(self, state, action, t=None):
        raise NotImplementedError

    def running_cost(self, state, action, t=None):
        raise NotImplementedError

    def terminal_state_cost(self, states, actions):
        raise NotImplementedError(f"{type(self).__name__} has no terminal cost")

    def __call__(self, state, action):
        return self.dynamics(state, action)

    # -- device-side parameters ----------------------------------------------------------------
    def _param_list(self):
        return []

    def flags(self):
        """MPPI_MODEL_FLAG_* for the problem block (include/mppi_amd.h)"""
        return 0

    def param_blob(self, device, dtype):
        """Flat parameter vector in the layout csrc/models.hpp documents (cached per device/dtype)."""
        key = (str(device), dtype)
        if key not in self._blob_cache:
            parts = [torch.as_tensor(p, dtype=dtype).reshape(-1) for p in self._param_list()]
            blob = torch.cat(parts) if parts else torch.zeros(1, dtype=dtype)
            self._blob_cache[key] = blob.to(device).contiguous()
        return self._blob_cache[key]

    def invalidate(self):
        """Call after mutating parameters in place (e.g. re-trained MLP weights)."""
        self._blob_cache.clear()
        self._param_version += 1


class Pendulum(NativeModel):
    """gym Pendulum-v1 true dynamics + cost, /root/reference/tests/pendulum.py:30-60."""
    model_id = N.MODEL_PENDULUM
    nx, nu = 2, 1

    def dynamics(self, state, action, t=None):
        th = state[:, 0:1]
        thdot = state[:, 1:2]
        u = torch.clamp(action, -2, 2)                                   # pendulum.py:41-42
        newthdot = thdot + (15.0 * torch.sin(th) + 3.0 * u) * 0.05       # :44
        newthdot = torch.clamp(newthdot, -8, 8)                          # :45
        newth = th + newthdot * 0.05                                     # :46
        return self._noisy(torch.cat((newth, newthdot), dim=1))

    def running_cost(self, state, action, t=None):
        an = ((state[:, 0] + math.pi) % (2 * math.pi)) - math.pi         # :52-53
        return an ** 2 + 0.1 * state[:, 1] ** 2                          # :56-61


class Integrator(NativeModel):
    """n-D integrator ("quad-toy"): x[:nu] += u, cost = sum x^2
    (/root/reference/tests/benchmark_mppi.py:65-78)."""
    model_id = N.MODEL_INTEGRATOR

    def __init__(self, nx, nu):
        super().__init__()
        self.nx, self.nu = int(nx), int(nu)

    def dynamics(self, state, action, t=None):
        nxt = state.clone()
        nxt[..., :self.nu] = nxt[..., :self.nu] + action
        return self._noisy(nxt)

    def running_cost(self, state, action, t=None):
        return (state ** 2).sum(dim=-1)


class LinearGoal(NativeModel):
    """x' = x + u @ B.T, cost = sum((goal-x)^2), terminal = the same on the last state
    (/root/reference/tests/test_mppi.py:25-51)."""
    model_id = N.MODEL_LINEAR_GOAL
    has_terminal = True

    def __init__(self, B, goal):
        super().__init__()
        self.B = torch.as_tensor(B)
        self.goal = torch.as_tensor(goal)
        self.nx, self.nu = self.B.shape

    def _param_list(self):
        return [self.B, self.goal]

    def _on(self, ref):
        return self.B.to(ref.device, ref.dtype), self.goal.to(ref.device, ref.dtype)

    def dynamics(self, state, action, t=None):
        B, _ = self._on(state)
        return self._noisy(state + action @ B.T)

    def running_cost(self, state, action, t=None):
        _, goal = self._on(state)
        return ((goal - state) ** 2).sum(dim=-1)

    def terminal_state_cost(self, states, actions):
        _, goal = self._on(states)
        return ((goal - states[..., -1, :]) ** 2).sum(dim=-1)


# (nx, nu) with an instantiation of the split-operand matrix-core kernel (csrc/rollout_mlp_split.hip MPPI_SPLIT_DIMS_*)
MLP_MATRIX_CORE_SHAPES = ((16, 4), (8, 2), (12, 6), (16, 8))


def mlp_kernel_width(nx, nu, hidden):
    """The hidden width the engine's matrix-core MLP kernels are built for (csrc/rollout_mlp_split.hip: the (nx, nu) of
    MLP_MATRIX_CORE_SHAPES, hidden 64 / 128 / 256) that holds `hidden` units, or `hidden` itself where no such kernel
    exists (the per-lane kernel takes any width).  Padding units have zero weights in and out and zero bias: tanh(0) = 0
    contributes exactly
    nothing."""
    if (int(nx), int(nu)) in MLP_MATRIX_CORE_SHAPES:
        for w in (64, 128, 256):
            if hidden <= w:
                return w
    return int(hidden)


def pad_hidden(W1, b1, W2, width):
    """(W1 (H, ni), b1 (H), W2 (nx, H)) with the hidden axis zero-padded to `width` units (float64 copies)"""
    W1, b1, W2 = (torch.as_tensor(t).detach().to("cpu", torch.float64) for t in (W1, b1, W2))
    H = W1.shape[0]
    if width == H:
        return W1, b1, W2
    W1p = torch.zeros(width, W1.shape[1], dtype=torch.float64)
    W1p[:H] = W1
    b1p = torch.zeros(width, dtype=torch.float64)
    b1p[:H] = b1
    W2p = torch.zeros(W2.shape[0], width, dtype=torch.float64)
    W2p[:, :H] = W2
    return W1p, b1p, W2p


class MLPResidual(NativeModel):
    """x' = x + res_scale * (W2 tanh(W1 [x;u] + b1) + b2) -- the 2-layer approximate-dynamics shape of
    /root/reference/tests/pendulum_approximate.py:47-67 -- with the diagonal quadratic running cost
    sum_i q_state[i] x_i^2 + sum_n q_control[n] u_n^2 (defaults 1 and 0: the plain sum x^2 of BASELINE
    configs[3..4])."""
    model_id = N.MODEL_MLP

    def __init__(self, W1, b1, W2, b2, nx, nu, res_scale=0.1, q_state=None, q_control=None):
        super().__init__()
        self.W1, self.b1, self.W2, self.b2 = (torch.as_tensor(t) for t in (W1, b1, W2, b2))
        self.nx, self.nu = int(nx), int(nu)
        # `hidden` is what the kernels see: odd widths are zero-padded to the next width the matrix-core kernels are
        # built for (hidden 100 on the per-lane kernel costs 13 x the matrix-core time)
        self.hidden_units = int(self.W1.shape[0])
        self.hidden = mlp_kernel_width(nx, nu, self.hidden_units)
        self.res_scale = float(res_scale)
        self.q_state = torch.ones(self.nx, dtype=torch.float64) if q_state is None else torch.as_tensor(q_state,
                dtype=torch.float64).reshape(-1)
        self.q_control = torch.zeros(self.nu, dtype=torch.float64) if q_control is None else torch.as_tensor(q_control,
                dtype=torch.float64).reshape(-1)
        assert self.W1.shape == (self.hidden_units, self.nx + self.nu) and self.W2.shape == (self.nx, self.hidden_units)
        assert self.q_state.numel() == self.nx and self.q_control.numel() == self.nu
        self._plain_cost = q_state is None and q_control is None

    def flags(self):
        """The default matrix-core kernel runs layer 2 on two-piece fp16 operands (csrc/rollout_mlp_split.hip): its
        weights (-2 W2) must stay inside fp16's range.  Weights beyond it -- checked here every time the parameter
        version changes, i.e. also after `invalidate()` behind an in-place update -- select the exact fp32 MFMA kernel,
        which has no such
        limit."""
        if getattr(self, "_flags_version", None) != self._param_version:
            self._flags = N.MODEL_FLAG_EXACT_FP32 if float(torch.as_tensor(self.W2).abs().max()) >= 3.0e4 else 0
            self._flags_version = self._param_version
        return self._flags

    @classmethod
    def random(cls, nx, nu, hidden, seed=2, dtype=torch.float32, res_scale=0.1):
        """torch.nn.Linear default initialisation under a private generator state."""
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        l1 = torch.nn.Linear(nx + nu, hidden)
        l2 = torch.nn.Linear(hidden, nx)
        torch.random.set_rng_state(g)
        with torch.no_grad():
            return cls(l1.weight.detach().to(dtype).clone(), l1.bias.detach().to(dtype).clone(),
                       l2.weight.detach().to(dtype).clone(), l2.bias.detach().to(dtype).clone(),
                       nx, nu, res_scale)

    def _param_list(self):
        W1, b1, W2 = (self.W1, self.b1, self.W2) if self.hidden == self.hidden_units else pad_hidden(self.W1, self.b1,
                self.W2, self.hidden)
        return [W1, b1, W2, self.b2, torch.tensor([self.res_scale], dtype=torch.float64), self.q_state, self.q_control]

    def dynamics(self, state, action, t=None):
        W1, b1, W2, b2 = (t.to(state.device, state.dtype) for t in (self.W1, self.b1, self.W2, self.b2))
        h = torch.tanh(torch.cat((state, action), dim=1) @ W1.T + b1)
        return self._noisy(state + self.res_scale * (h @ W2.T + b2))

    def running_cost(self, state, action, t=None):
        if self._plain_cost:
            return (state ** 2).sum(dim=-1)
        return ((self.q_state.to(state.device, state.dtype) * state ** 2).sum(dim=-1)
                + (self.q_control.to(state.device, state.dtype) * action ** 2).sum(dim=-1))


def native_model_of(dynamics, running_cost, terminal_state_cost=None):
    """The NativeModel whose bound methods were passed as the reference-style callables, or None."""
    m = dynamics if isinstance(dynamics, NativeModel) else getattr(dynamics, "__self__", None)
    if not isinstance(m, NativeModel):
        return None
    if getattr(dynamics, "__func__", None) not in (None, type(m).dynamics) and not isinstance(dynamics, NativeModel):
        return None
    rc_self = getattr(running_cost, "__self__", None)
    if rc_self is not m or getattr(running_cost, "__func__", None) is not type(m).running_cost:
        return None
    if terminal_state_cost is not None:
        ts = getattr(terminal_state_cost, "__self__", None)
        if ts is not m or getattr(terminal_state_cost, "__func__", None) is not type(m).terminal_state_cost:
            return None
        if not m.has_terminal:
            return None
    return m
