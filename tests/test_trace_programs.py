"""pytorch_mppi_amd/trace.py on whole programs: dynamics / cost pairs the way people write them for MPPI (classic control
equations, a class with obstacle loops and a terminal cost, einsum quadratic forms, RK4 and sub-stepping loops, quaternions,
nn.Modules with buffers / LayerNorm / ensembles, schedules indexed by the timestep).  Each must translate and agree with the
callable on random batches (host build, fp64); the one idiom that cannot be traced (a Python list indexed by the timestep)
must be refused, not mistranslated."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from pytorch_mppi_amd import trace

PROGRAMS = {}


def check(name, f, q, nx, nu, term=None, sd=False, horizon=None):
    PROGRAMS[name] = (f, q, nx, nu, term, sd, horizon)


# 1. cartpole (classic control equations)
def cartpole(state, action):
    g, mc, mp, l, dt = 9.8, 1.0, 0.1, 0.5, 0.02
    x, xd, th, thd = state[:, 0], state[:, 1], state[:, 2], state[:, 3]
    force = torch.clamp(action[:, 0], -10, 10)
    costh, sinth = torch.cos(th), torch.sin(th)
    temp = (force + mp * l * thd ** 2 * sinth) / (mc + mp)
    thacc = (g * sinth - costh * temp) / (l * (4.0 / 3.0 - mp * costh ** 2 / (mc + mp)))
    xacc = temp - mp * l * thacc * costh / (mc + mp)
    return torch.stack((x + dt * xd, xd + dt * xacc, th + dt * thd, thd + dt * thacc), dim=1)
def cartpole_cost(state, action):
    return state[:, 0] ** 2 + 10 * (1 - torch.cos(state[:, 2])) + 0.1 * state[:, 1] ** 2 + 0.1 * state[:, 3] ** 2 + 0.001 * action[:, 0] ** 2
check('cartpole', cartpole, cartpole_cost, 4, 1)

# 2. Dubins car with obstacles, class-based
class Dubins:
    def __init__(self):
        self.dt = 0.1
        self.goal = torch.tensor([3.0, 3.0])
        self.obs = torch.tensor([[1.0, 1.0, 0.5], [2.0, 2.5, 0.4]])
    def dynamics(self, state, u):
        x, y, th = state[:, 0], state[:, 1], state[:, 2]
        v = torch.clamp(u[:, 0], 0.0, 1.5); w = torch.clamp(u[:, 1], -1.0, 1.0)
        ns = torch.zeros_like(state)
        ns[:, 0] = x + v * torch.cos(th) * self.dt
        ns[:, 1] = y + v * torch.sin(th) * self.dt
        ns[:, 2] = th + w * self.dt
        return ns
    def cost(self, state, u):
        pos = state[:, :2]
        c = torch.norm(pos - self.goal.to(state.device, state.dtype), dim=1)
        for o in self.obs:
            d = torch.norm(pos - o[:2].to(state.device, state.dtype), dim=1)
            c = c + 1000.0 * (d < o[2]).float()
        return c
    def terminal(self, states, actions):
        return 10.0 * torch.norm(states[..., -1, :2] - self.goal.to(states.device, states.dtype), dim=-1)
d = Dubins()
check('dubins_obstacles_class', d.dynamics, d.cost, 3, 2, d.terminal)

# 3. planar quadrotor
def quad2d(s, u):
    m, I, r, g, dt = 0.5, 0.01, 0.2, 9.81, 0.02
    x, y, th, vx, vy, om = s.unbind(1)
    u1, u2 = u[:, 0].clamp(0, 10), u[:, 1].clamp(0, 10)
    ax = -(u1 + u2) * torch.sin(th) / m
    ay = (u1 + u2) * torch.cos(th) / m - g
    al = r * (u1 - u2) / I
    return torch.stack([x + dt * vx, y + dt * vy, th + dt * om, vx + dt * ax, vy + dt * ay, om + dt * al], -1)
Qm = torch.diag(torch.tensor([10., 10., 1., 1., 1., 0.1]))
def quad_cost(s, u):
    e = s - torch.tensor([1., 1., 0, 0, 0, 0], dtype=s.dtype, device=s.device)
    return torch.einsum('bi,ij,bj->b', e, Qm.to(s.dtype), e) + 0.01 * (u ** 2).sum(1)
check('quad2d_einsum_cost', quad2d, quad_cost, 6, 2)

# 4. double integrator with A, B matrices and x @ A.T
A = torch.tensor([[1, 0, 0.1, 0], [0, 1, 0, 0.1], [0, 0, 1, 0], [0, 0, 0, 1.0]])
B = torch.tensor([[0.005, 0], [0, 0.005], [0.1, 0], [0, 0.1]])
check('double_integrator_AB', lambda x, u: x @ A.T.to(x.dtype) + u @ B.T.to(x.dtype), lambda x, u: (x[:, :2] ** 2).sum(1) + 1e-3 * (u * u).sum(1), 4, 2)

# 5. kinematic bicycle with atan / tan
def bicycle(s, u):
    L, dt = 2.5, 0.05
    x, y, psi, v = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    a, delta = u[:, 0], torch.clamp(u[:, 1], -0.5, 0.5)
    beta = torch.atan(0.5 * torch.tan(delta))
    return torch.stack((x + dt * v * torch.cos(psi + beta), y + dt * v * torch.sin(psi + beta), psi + dt * v / (0.5 * L) * torch.sin(beta), v + dt * a), 1)
def bicycle_cost(s, u):
    lat = s[:, 1] - 0.5 * torch.sin(0.3 * s[:, 0])
    return lat ** 2 + (s[:, 3] - 5.0) ** 2 * 0.1 + torch.relu(torch.abs(s[:, 1]) - 2.0) * 100
check('bicycle_atan_tan', bicycle, bicycle_cost, 4, 2)

# 6. time-varying reference tracking with step_dependent
ref = torch.sin(torch.linspace(0, 3, 50))
def track_cost(s, u, t):
    return (s[:, 0] - ref[t]) ** 2
check('tracking_ref_indexed_by_t', lambda s, u, t: s + 0.1 * u, track_cost, 2, 2, None, True, 50)

# 7. numpy-style pendulum with in-place angle normalize function and np.pi
def angle_normalize(x): return (((x + np.pi) % (2 * np.pi)) - np.pi)
def pend(state, action):
    th = state[:, 0].view(-1, 1); thdot = state[:, 1].view(-1, 1)
    u = torch.clamp(action, -2, 2)
    newthdot = thdot + (-3 * 10 / 2 * torch.sin(th + np.pi) + 3. * u) * 0.05
    newth = th + newthdot * 0.05
    newthdot = torch.clamp(newthdot, -8, 8)
    return torch.cat((newth, newthdot), dim=1)
check('pendulum_np_pi', pend, lambda s, a: angle_normalize(s[:, 0]) ** 2 + .1 * s[:, 1] ** 2 + .001 * a[:, 0] ** 2, 2, 1)

# 8. GRU-less small residual net with LayerNorm + skip, state normalisation buffers
import torch.nn as nn
class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('mu', torch.tensor([0.1, -0.1, 0.0, 0.2, 0., 0.])); self.register_buffer('sd', torch.tensor([1., 2., 1., .5, 1., 1.]))
        self.l1 = nn.Linear(6, 16); self.ln = nn.LayerNorm(16); self.l2 = nn.Linear(16, 16); self.l3 = nn.Linear(16, 4)
    def forward(self, s, a):
        z = (torch.cat((s, a), -1) - self.mu) / self.sd
        h = torch.relu(self.ln(self.l1(z)))
        h = h + torch.relu(self.l2(h))
        return s + self.l3(h)
net = Net().double()
check('residual_net_layernorm_buffers', lambda s, a: net(s, a), lambda s, a: (s ** 2).sum(-1), 4, 2)

# 1. dataclass-ish config, dict params, python loops over dims
class Cfg: dt = 0.05; mass = 1.2; drag = torch.tensor([0.1, 0.2, 0.3])
def pm3(s, u):
    p, v = s[:, :3], s[:, 3:]
    a = (u - Cfg.drag.to(s.device) * v * v.abs()) / Cfg.mass
    a = a + torch.tensor([0.0, 0.0, -9.81], device=s.device, dtype=s.dtype)
    v2 = v + Cfg.dt * a
    return torch.cat([p + Cfg.dt * v2, v2], dim=-1)
check('point_mass_3d_drag', pm3, lambda s, u: sum((s[:, i] - g) ** 2 for i, g in enumerate([1.0, 2.0, 3.0])) + 1e-3 * u.pow(2).sum(-1), 6, 3)

# 2. torch.sqrt + eps, division by norm, torch.where with tensors, soft constraints with relu / softplus / exp barrier
def swim(s, u):
    sp = torch.sqrt((s[:, 2:4] ** 2).sum(-1, keepdim=True) + 1e-8)
    dirn = s[:, 2:4] / sp
    acc = u - 0.5 * sp * dirn
    v = s[:, 2:4] + 0.1 * acc
    return torch.cat((s[:, :2] + 0.1 * v, v), 1)
def swim_cost(s, u):
    d = (s[:, :2] - torch.tensor([2.0, 0.0], dtype=s.dtype, device=s.device)).norm(dim=-1)
    wall = F.softplus(10 * (s[:, 1].abs() - 1.0)) + torch.exp(5 * (s[:, 0] - 3.0)).clamp(max=1e3)
    return d + wall + torch.where(d < 0.1, torch.zeros_like(d), 0.1 * torch.ones_like(d))
check('swimmer_barriers', swim, swim_cost, 4, 2)

# 3. Runge-Kutta 4 integration of a nested function (van der pol)
def vdp(x, u):
    return torch.stack((x[:, 1], 1.5 * (1 - x[:, 0] ** 2) * x[:, 1] - x[:, 0] + u[:, 0]), 1)
def rk4(s, u, h=0.05):
    k1 = vdp(s, u); k2 = vdp(s + 0.5 * h * k1, u); k3 = vdp(s + 0.5 * h * k2, u); k4 = vdp(s + h * k3, u)
    return s + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
check('rk4_van_der_pol', rk4, lambda s, u: (s ** 2).sum(1), 2, 1)

# 4. multi-step substepping loop
def substeps(s, u):
    for _ in range(4):
        s = s + 0.025 * torch.cat((s[:, 1:2], -torch.sin(s[:, 0:1]) + u), 1)
    return s
check('substep_loop', substeps, lambda s, u: 1 - torch.cos(s[:, 0]) + 0.1 * s[:, 1] ** 2, 2, 1)

# 5. quaternion attitude kinematics (normalize, hamilton product via stack)
def quat(s, w):
    q0, q1, q2, q3 = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    wx, wy, wz = w[:, 0], w[:, 1], w[:, 2]
    dq = 0.5 * torch.stack((-q1 * wx - q2 * wy - q3 * wz, q0 * wx + q2 * wz - q3 * wy, q0 * wy - q1 * wz + q3 * wx, q0 * wz + q1 * wy - q2 * wx), -1)
    q = s + 0.02 * dq
    return q / q.norm(dim=-1, keepdim=True)
check('quaternion', quat, lambda s, w: 1 - s[:, 0] ** 2 + 0.01 * (w ** 2).sum(-1), 4, 3)

# 6. ensemble of two small nets averaged (ModuleList), with eval() + no_grad inside
nets = nn.ModuleList([nn.Sequential(nn.Linear(3, 8), nn.SiLU(), nn.Linear(8, 2)) for _ in range(2)]).double().eval()
def ens(s, a):
    with torch.no_grad():
        xu = torch.cat((s, a), -1)
        return s + torch.stack([n(xu) for n in nets]).mean(0)
check('ensemble_mean', ens, lambda s, a: (s ** 2).sum(-1), 2, 1)

# 7. terminal cost using final state & goal, running cost with action rate via t (step dependent weights list)
w = [1.0, 0.9, 0.8, 0.7, 0.6, 0.5]
check('python_list_indexed_by_t', lambda s, a, t: s + 0.1 * a, lambda s, a, t: w[t] * (s ** 2).sum(-1), 2, 2, None, True, 6)

# 8. cost using torch.max over a stacked tensor of penalties + min over obstacles
def pen(s, a):
    cands = torch.stack((s[:, 0].abs() - 1, s[:, 1].abs() - 2, a[:, 0].abs() - 0.5), 1)
    return torch.max(cands, dim=1)[0].clamp(min=0) * 50 + torch.min(torch.stack(((s[:, 0] - 1) ** 2, (s[:, 0] + 1) ** 2), 1), dim=1).values
check('max_min_penalties', lambda s, a: s + 0.1 * a, pen, 2, 2)

# 9. dynamics with matrix exponential-free linearisation x' = x + dt (A x + B u) using torch.matmul with batch dims
A2 = torch.tensor([[0., 1.], [-1., -0.1]]); B2 = torch.tensor([[0.], [1.]])
check('matmul_batched_col_vectors', lambda x, u: x + 0.05 * (torch.matmul(A2.to(x.dtype), x.unsqueeze(-1)) + torch.matmul(B2.to(x.dtype), u.unsqueeze(-1))).squeeze(-1), lambda x, u: (x * x).sum(-1), 2, 1)

# 10. integer control via rounding / sign (bang-bang)
check('bang_bang_sign_round', lambda s, u: s + 0.1 * torch.sign(u) + 0.01 * torch.round(u * 2) / 2, lambda s, u: (s ** 2).sum(-1), 2, 2)


UNTRACEABLE = {"python_list_indexed_by_t"}


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_program(name):
    f, q, nx, nu, term, sd, horizon = PROGRAMS[name]
    if name in UNTRACEABLE:
        with pytest.raises(trace.TraceUnsupported):
            trace.generate(f, q, nx, nu, term, sd)
        return
    code = trace.generate(f, q, nx, nu, term, sd)
    assert trace.verify_on_host(code, f, q, nx, nu, term, sd, horizon=horizon)
