"""User models used by tests/test_gpu_jit_models.py.  Defined here (not in the test) so that
`__graft_entry__.build()` can compile them in the build container: the objects land in
pytorch_mppi_amd/_jit/ (in-tree, hash-named) and travel to the GPU box with the snapshot."""
import math

import torch

import pytorch_mppi_amd as pm
from pytorch_mppi_amd import jit

DT_, GX, GY, WT = 0.1, 1.5, -0.5, 3.0


def pendulum_user():
    builtin = pm.models.Pendulum()
    return builtin, jit.compile_model(
        "pendulum_user", 2, 1, dynamics=builtin.dynamics, running_cost=builtin.running_cost,
        step="const T uc = clampT(u[0], T(-2), T(2));"
             "T nthd = x[1] + (T(15) * m_sin_moderate(x[0]) + T(3) * uc) * T(0.05);"
             "nthd = clampT(nthd, T(-8), T(8)); x[0] = x[0] + nthd * T(0.05); x[1] = nthd;",
        cost="const T pi = T(3.141592653589793), two_pi = T(6.283185307179586);"
             "const T an = m_floormod(x[0] + pi, two_pi) - pi; return an * an + T(0.1) * (x[1] * x[1]);")


def unicycle_callables():
    def f(s, a):
        return torch.stack((s[:, 0] + DT_ * a[:, 0] * torch.cos(s[:, 2]), s[:, 1] + DT_ * a[:, 0] * torch.sin(s[:, 2]),
                            s[:, 2] + DT_ * a[:, 1]), dim=1)

    def q(s, a):
        return (s[:, 0] - GX) ** 2 + (s[:, 1] - GY) ** 2 + 0.01 * (a ** 2).sum(-1)

    def term(states, actions):
        last = states[..., -1, :]
        return WT * ((last[..., 0] - GX) ** 2 + (last[..., 1] - GY) ** 2)

    return f, q, term


def unicycle():
    f, q, term = unicycle_callables()
    return jit.compile_model(
        "unicycle", 3, 2, dynamics=f, running_cost=q, terminal_state_cost=term, params=[DT_, GX, GY, WT],
        step="const T c = m_cos(x[2]), s = m_sin(x[2]); x[0] += p[0] * u[0] * c; x[1] += p[0] * u[0] * s; x[2] += p[0] * u[1];",
        cost="const T dx = x[0] - p[1], dy = x[1] - p[2]; return dx * dx + dy * dy + T(0.01) * (u[0] * u[0] + u[1] * u[1]);",
        terminal="const T dx = x[0] - p[1], dy = x[1] - p[2]; return p[3] * (dx * dx + dy * dy);")


def drifting_callables(dtype=torch.float64):
    """time-VARYING dynamics and cost (step_dependent_dynamics=True, mppi.py:147-154): the control authority
    grows with the timestep, the goal moves"""
    def f(s, a, t):
        return s + DT_ * (1.0 + 0.05 * t) * a

    def q(s, a, t):
        return (s[:, 0] - 0.1 * t) ** 2 + (s[:, 1] + 0.05 * t) ** 2 + 0.01 * (a ** 2).sum(-1)

    return f, q


def drifting():
    f, q = drifting_callables()
    return jit.compile_model(
        "drifting", 2, 2, dynamics=f, running_cost=q, params=[DT_], step_dependent=True,
        step="const T g = p[0] * (T(1) + T(0.05) * T(t)); x[0] += g * u[0]; x[1] += g * u[1];",
        cost="const T dx = x[0] - T(0.1) * T(t), dy = x[1] + T(0.05) * T(t); return dx * dx + dy * dy + T(0.01) * (u[0] * u[0] + u[1] * u[1]);")


def cart4_callables():
    """4 controls, 6 states: damped double integrator in three axes with a coupled yaw control -- a user model
    with nu % 4 == 0, i.e. one the KMPPI-fused K1 (interpolation inside the kernel) is instantiated for"""
    def f(s, a):
        v = 0.9 * s[:, 3:6] + DT_ * a[:, 0:3] + 0.02 * a[:, 3:4] * s[:, [1, 2, 0]]
        return torch.cat((s[:, 0:3] + DT_ * v, v), dim=1)

    def q(s, a):
        return ((s[:, 0] - GX) ** 2 + (s[:, 1] - GY) ** 2 + s[:, 2] ** 2 + 0.1 * (s[:, 3:6] ** 2).sum(-1)
                + 0.01 * (a ** 2).sum(-1))

    return f, q


def cart4():
    f, q = cart4_callables()
    return jit.compile_model(
        "cart4", 6, 4, dynamics=f, running_cost=q, params=[DT_, GX, GY],
        step="const T v0 = T(0.9) * x[3] + p[0] * u[0] + T(0.02) * u[3] * x[1];"
             "const T v1 = T(0.9) * x[4] + p[0] * u[1] + T(0.02) * u[3] * x[2];"
             "const T v2 = T(0.9) * x[5] + p[0] * u[2] + T(0.02) * u[3] * x[0];"
             "x[0] += p[0] * v0; x[1] += p[0] * v1; x[2] += p[0] * v2; x[3] = v0; x[4] = v1; x[5] = v2;",
        cost="const T dx = x[0] - p[1], dy = x[1] - p[2];"
             "return dx * dx + dy * dy + x[2] * x[2] + T(0.1) * (x[3] * x[3] + x[4] * x[4] + x[5] * x[5])"
             " + T(0.01) * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);")


# ---- plain torch callables for jit.from_torch (tests/test_trace.py, tests/test_gpu_from_torch.py) ----------------------
# restated after the reference's own test callables: the gym pendulum with numpy ufuncs applied to tensors
# (/root/reference/tests/pendulum.py:30-60) and the linear dynamics + quadratic goal cost with a terminal cost
# (/root/reference/tests/test_mppi.py:25-51)
def ref_pendulum_callables():
    import math
    import numpy as np

    def dynamics(state, perturbed_action):
        th = state[:, 0].view(-1, 1)
        thdot = state[:, 1].view(-1, 1)
        g, m, l, dt = 10, 1, 1, 0.05
        u = torch.clamp(perturbed_action, -2, 2)
        newthdot = thdot + (3 * g / (2 * l) * np.sin(th) + 3.0 / (m * l ** 2) * u) * dt
        newthdot = np.clip(newthdot, -8, 8)
        newth = th + newthdot * dt
        return torch.cat((newth, newthdot), dim=1)

    def angle_normalize(x):
        return ((x + math.pi) % (2 * math.pi)) - math.pi

    def running_cost(state, action):
        theta, theta_dt = state[:, 0], state[:, 1]
        return angle_normalize(theta) ** 2 + 0.1 * theta_dt ** 2

    return dynamics, running_cost


def ref_linear_callables(dtype=torch.float64):
    B = torch.tensor([[1.0, 0.0], [0.0, -1.0]], dtype=dtype)
    goal = torch.tensor([2.0, 2.0], dtype=dtype)

    def dynamics(state, action):
        return state + action @ B.to(state.device, state.dtype).T

    def cost(state, action):
        dx = goal.to(state.device, state.dtype) - state
        return (dx ** 2).sum(dim=-1)

    def terminal(states, actions):
        dx = goal.to(states.device, states.dtype) - states[..., -1, :]
        return (dx ** 2).sum(dim=-1)

    return dynamics, cost, terminal


def small_mlp_callables(nx=4, nu=2, hidden=8, seed=5):
    """an nn.Module as dynamics: Linear -> Tanh -> Linear residual (the shape of /root/reference/tests/pendulum_approximate.py:47-67)"""
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(nx + nu, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, nx)).double()
    for p_ in net.parameters():
        p_.requires_grad_(False)

    def dynamics(state, action):
        return state + 0.1 * net.to(state.device, state.dtype)(torch.cat((state, action), dim=1))

    def cost(state, action):
        return (state ** 2).sum(dim=1) + 0.05 * torch.abs(action).sum(dim=1)

    return dynamics, cost


def approx_pendulum_callables(hidden=32, seed=25, dtype=torch.double):
    """learned dynamics in the shape of /root/reference/tests/pendulum_approximate.py:47-67,98-108: a TRAINABLE residual
    network (nx+nu -> hidden -> hidden -> nx, tanh) whose parameters the user's training loop keeps writing between
    commands; the action clamped, the angle wrapped after the step.  Returns (dynamics, running_cost, network)."""
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(3, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, hidden), torch.nn.Tanh(),
                              torch.nn.Linear(hidden, 2)).to(dtype)
    wrap = lambda a: ((a + math.pi) % (2 * math.pi)) - math.pi

    def dynamics(state, perturbed_action):
        u = torch.clamp(perturbed_action, -2.0, 2.0)
        nxt = state + net(torch.cat((state, u), dim=1))
        nxt[:, 0] = wrap(nxt[:, 0])
        return nxt

    def cost(state, action):
        return wrap(state[:, 0]) ** 2 + 0.1 * state[:, 1] ** 2

    return dynamics, cost, net


def zoo_callables(seed=3):
    """a network with the activations users actually pick (GELU -> erf on the device, ELU -> expm1, ReLU) and a cost made
    of a Huber term, a norm and a logical mask: the wider operator vocabulary of the tracer on the GPU"""
    import torch.nn.functional as F
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.GELU(), torch.nn.Linear(8, 8), torch.nn.ELU(), torch.nn.Linear(8, 8),
                              torch.nn.ReLU(), torch.nn.Linear(8, 4)).double()
    goal = torch.tensor([0.5, -0.25, 0.0, 1.0], dtype=torch.float64)

    def dynamics(state, action):
        nxt = state + 0.1 * net.to(state.device)(torch.cat((state, action), dim=1))
        nxt[:, 3] = torch.atan(nxt[:, 3])
        return nxt

    def cost(state, action):
        g_ = goal.to(state.device)
        return (F.smooth_l1_loss(state, g_.expand_as(state), reduction="none", beta=0.5).sum(1) + 0.1 * torch.linalg.norm(action, dim=1)
                + torch.where((state[:, 0] > 0.5) & (state[:, 1] < 0.0), 1.0, 0.0))

    return dynamics, cost, net


def relu_net_callables(seed=7, dtype=torch.float32):
    """nx = 4, nu = 2 dynamics network with the activations that are NOT all one-operand functions of their layer (GELU reads its
    argument twice: that layer stands alone; ReLU and a scaled sigmoid fuse) and an output wider than the state"""
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(6, 24), torch.nn.GELU(), torch.nn.Linear(24, 24), torch.nn.ReLU(), torch.nn.Linear(24, 20),
                              torch.nn.Sigmoid(), torch.nn.Linear(20, 6)).to(dtype)

    def dynamics(state, action):
        out = net(torch.cat((state, action), dim=1))
        return state + 0.1 * out[:, :4] * (1.0 + 0.1 * out[:, 4:5])

    def cost(state, action):
        return (state ** 2).sum(dim=1) + 0.05 * (action ** 2).sum(dim=1)

    return dynamics, cost, net


def tracking_callables(T=24):
    """time-varying reference (step_dependent_dynamics=True): a schedule tensor indexed by the timestep, `ref[t]` / `ref[t + 1]`
    -- one small constant table per looked-up element in the traced functor"""
    ref = torch.stack((torch.sin(torch.linspace(0, 3, T + 1)), torch.cos(torch.linspace(0, 2, T + 1))), 1).double()     # (T + 1, 2)
    gain = torch.linspace(1.0, 0.5, T).double()

    def dynamics(state, action, t):
        return state + 0.1 * action * gain.to(state.device)[t]

    def cost(state, action, t):
        r = ref.to(state.device)
        return ((state - r[t]) ** 2).sum(-1) + 0.1 * (state[:, 0] - r[t + 1, 0]) ** 2 + 0.01 * (action ** 2).sum(-1)

    return dynamics, cost


def approx_terminal_cost(states, actions):
    """terminal cost used with the learned pendulum in tests/test_gpu_from_torch.py (its traced form is built by build())"""
    return 3.0 * (states[..., -1, :] ** 2).sum(-1)


def train_a_little(net, steps=3, seed=0):
    """a few optimizer steps on random targets: what happens to the network between two commands"""
    g = torch.Generator().manual_seed(seed)
    p0 = next(net.parameters())
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    for _ in range(steps):
        xu = torch.randn(64, 3, generator=g, dtype=torch.float64).to(p0.device, p0.dtype)
        y = torch.randn(64, 2, generator=g, dtype=torch.float64).to(p0.device, p0.dtype)
        opt.zero_grad()
        ((net(xu) - y) ** 2).mean().backward()
        opt.step()


def watched_linear_callables():
    """a constant matrix the test later writes in place (tests/test_gpu_from_torch.py: the controller must notice)"""
    B = torch.tensor([[1.0, 0.0], [0.0, -1.0]], dtype=torch.float64)
    return (lambda s, a: s + a @ B.to(s.device).T), (lambda s, a: (s ** 2).sum(-1)), B


class RetargetedCost:
    """state the reference reads live on every call (/root/reference/tests/smooth_mppi.py:54-58: `dx = self.goal - state`):
    a goal TENSOR the user rebinds between commands and a Python float gain"""
    def __init__(self):
        self.goal = torch.tensor([1.5, -0.5], dtype=torch.float64)
        self.gain = 2.0
        self.calls = 0

    def __call__(self, state, action):
        self.calls += 1
        dx = self.goal.to(state.device, state.dtype) - state
        return self.gain * (dx ** 2).sum(-1) + 0.01 * (action ** 2).sum(-1)


def retargeted_callables():
    """(dynamics, cost object, holder): linear dynamics through a matrix in a holder dict, the cost above"""
    holder = {"B": torch.tensor([[0.2, 0.0], [0.05, -0.2]], dtype=torch.float64)}
    cost = RetargetedCost()

    def dynamics(state, action):
        return state + action @ holder["B"].to(state.device, state.dtype).T

    return dynamics, cost, holder


RETARGET_GOAL_2, RETARGET_GAIN_2 = [-1.0, 0.75], 3.0


def swappable_net_callables(seed=11, hidden=6):
    """an nn.Module the user REPLACES between commands (same architecture, other weights): holder['net'] = new_net"""
    mk = lambda s_: _frozen_mlp(s_, hidden)
    holder = {"net": mk(seed)}

    def dynamics(state, action):
        return state + 0.1 * holder["net"](torch.cat((state, action), dim=1).to(torch.float64)).to(state.dtype)

    def cost(state, action):
        return (state ** 2).sum(dim=1) + 0.05 * (action ** 2).sum(dim=1)

    return dynamics, cost, holder, mk


def _frozen_mlp(seed, hidden):
    g = torch.Generator().manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(3, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, 2)).double()
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g, dtype=torch.float64) * 0.4)
    return net


def promoted_paths(callables, tensors):
    """the places (watch.Path) at which `callables` read the given tensors: what the controller promotes to run-time
    parameters once it has seen them change (build() compiles those variants ahead, so that the GPU tests find them cached)"""
    from pytorch_mppi_amd import watch
    w = watch.StateWatch(list(callables))
    return [p_ for p_, kind, ref in w.places if kind == "t" and any(ref[0] is t for t in tensors)]


def traced_models():
    """the traced + compiled forms of the callables above (built by __graft_entry__.build() so that the objects travel)"""
    import concurrent.futures as cf
    f, q = ref_pendulum_callables()
    lf, lq, lt = ref_linear_callables()
    mf, mq = small_mlp_callables()
    wf, wq, _ = watched_linear_callables()
    af, aq, _ = approx_pendulum_callables()
    zf, zq, _ = zoo_callables()
    jobs = dict(pendulum=(f, q, 2, 1), linear=(lf, lq, 2, 2, lt), mlp=(mf, mq, 4, 2), watched=(wf, wq, 2, 2), approx=(af, aq, 2, 1),
                zoo=(zf, zq, 4, 2), approx_terminal=(af, aq, 2, 1, approx_terminal_cost))
    tf, tq = tracking_callables()
    # callables whose state the user changes between commands (tests/test_gpu_from_torch.py): the functor of the first trace,
    # and the one the controller compiles after it has seen the state move (the tensors that moved as run-time parameters)
    rf, rq, rh = retargeted_callables()
    jobs["retarget"] = (rf, rq, 2, 2)
    rf2, rq2, rh2 = retargeted_callables()
    rq2.gain = RETARGET_GAIN_2
    wf2, wq2, wB2 = watched_linear_callables()
    sf, sq, sh, _ = swappable_net_callables()
    jobs["swappable"] = (sf, sq, 2, 1)
    nf, nq, _ = relu_net_callables()
    jobs["relu_net"] = (nf, nq, 4, 2)
    af32, aq32, _ = approx_pendulum_callables(dtype=torch.float32)
    jobs["approx_f32"] = (af32, aq32, 2, 1)
    jobs["approx_f32_terminal"] = (af32, aq32, 2, 1, approx_terminal_cost)
    with cf.ThreadPoolExecutor(max_workers=8) as ex:      # each ends in its own hipcc subprocess
        futs = {k: ex.submit(jit.from_torch, *v) for k, v in jobs.items()}
        futs["tracking"] = ex.submit(jit.from_torch, tf, tq, 2, 2, step_dependent=True, horizon=24)
        futs["retarget_goal_promoted"] = ex.submit(jit.from_torch, rf2, rq2, 2, 2, dynamic=promoted_paths((rf2, rq2), [rq2.goal]))
        futs["retarget_goal_and_B_promoted"] = ex.submit(jit.from_torch, rf2, rq2, 2, 2,
                                                         dynamic=promoted_paths((rf2, rq2), [rq2.goal, rh2["B"]]))
        wf3, wq3, wB3 = watched_linear_callables()
        wB3.data[1, 0] = 0.75                             # (a write the version counters do not see: re-traced with the new constant)
        futs["watched_data_write"] = ex.submit(jit.from_torch, wf3, wq3, 2, 2)
        futs["watched_promoted"] = ex.submit(jit.from_torch, wf2, wq2, 2, 2, dynamic=promoted_paths((wf2, wq2), [wB2]))
        return {k: v.result() for k, v in futs.items()}
