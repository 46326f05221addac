"""TEST INFRASTRUCTURE ONLY -- torch restatements of the dynamics / cost callables that the
BASELINE.json configs and the reference's tests are quoted on.  They are ordinary
``dynamics(state, u)`` / ``running_cost(state, u)`` callables in the reference's plugin
convention (/root/reference/src/pytorch_mppi/mppi.py:63-64) and are what the oracle and the
live reference are driven with; the HIP device functors in
``pytorch_mppi_amd/csrc/models.hpp`` are independent implementations of the same formulas.

Nothing under ``pytorch_mppi_amd/`` imports this module.
"""
import math

import torch


# ---------------------------------------------------------------------------------------------
# Pendulum -- /root/reference/tests/pendulum.py:30-60 (gym Pendulum-v1 true dynamics).
# The reference writes np.sin / np.clip on tensors (tensor math); torch ops are the same maths.
# ---------------------------------------------------------------------------------------------
def pendulum_dynamics(state, perturbed_action):
    th = state[:, 0].view(-1, 1)
    thdot = state[:, 1].view(-1, 1)
    g, m, l, dt = 10, 1, 1, 0.05
    u = torch.clamp(perturbed_action, -2, 2)            # pendulum.py:41-42
    newthdot = thdot + (3 * g / (2 * l) * torch.sin(th) + 3.0 / (m * l ** 2) * u) * dt   # :44
    newthdot = torch.clamp(newthdot, -8, 8)             # :45
    newth = th + newthdot * dt                          # :46
    return torch.cat((newth, newthdot), dim=1)          # :48


def angle_normalize(x):
    return ((x + math.pi) % (2 * math.pi)) - math.pi    # pendulum.py:52-53


def pendulum_cost(state, action):
    theta = state[:, 0]
    theta_dt = state[:, 1]
    return angle_normalize(theta) ** 2 + 0.1 * theta_dt ** 2    # pendulum.py:56-61


# ---------------------------------------------------------------------------------------------
# "quad-toy" n-D integrator -- /root/reference/tests/benchmark_mppi.py:65-78
# ---------------------------------------------------------------------------------------------
def make_quadtoy(nx, nu):
    def dynamics(state, action):
        delta = torch.zeros_like(state)
        delta[..., :nu] = action                        # benchmark_mppi.py:67-69
        return state + delta

    def cost(state, action):
        return (state ** 2).sum(dim=-1)                 # benchmark_mppi.py:74-76

    return dynamics, cost


# ---------------------------------------------------------------------------------------------
# Linear dynamics + quadratic goal cost -- /root/reference/tests/test_mppi.py:25-51
# ---------------------------------------------------------------------------------------------
def make_linear_goal(B, goal):
    """x' = x + u @ B.T ; cost = sum((goal - x)^2) ; terminal = same on the last state."""

    def dynamics(state, action):
        return state + action @ B.T                     # test_mppi.py:28-29

    def cost(state, action):
        dx = goal - state
        return (dx ** 2).sum(dim=-1)                    # test_mppi.py:40-42

    def terminal(states, actions):
        dx = goal - states[..., -1, :]
        return (dx ** 2).sum(dim=-1)                    # test_mppi.py:49-51

    return dynamics, cost, terminal


# ---------------------------------------------------------------------------------------------
# 2-layer MLP residual dynamics (BASELINE.json configs[3..4]; shape after
# /root/reference/tests/pendulum_approximate.py:47-67: Linear -> Tanh -> Linear, state residual)
#   x' = x + res_scale * (W2 tanh(W1 [x;u] + b1) + b2),  cost = sum(x^2)
# Weights: torch.nn.Linear default init under torch.manual_seed(seed)  (SURVEY.md 8d).
# ---------------------------------------------------------------------------------------------
def make_mlp_weights(nx, nu, hidden, seed=2, dtype=torch.float32):
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    l1 = torch.nn.Linear(nx + nu, hidden)
    l2 = torch.nn.Linear(hidden, nx)
    torch.random.set_rng_state(g)
    with torch.no_grad():
        return (l1.weight.detach().to(dtype).clone(), l1.bias.detach().to(dtype).clone(),
                l2.weight.detach().to(dtype).clone(), l2.bias.detach().to(dtype).clone())


def make_mlp(W1, b1, W2, b2, res_scale=0.1, q_state=None, q_control=None):
    """q_state / q_control: the diagonal quadratic running cost sum q_i x_i^2 + sum r_n u_n^2 (None: the plain sum x^2)"""
    def dynamics(state, action):
        xu = torch.cat((state, action), dim=1)
        h = torch.tanh(xu @ W1.T + b1)
        return state + res_scale * (h @ W2.T + b2)

    def cost(state, action):
        if q_state is None and q_control is None:
            return (state ** 2).sum(dim=-1)
        c = ((state ** 2) * (1.0 if q_state is None else q_state.to(state))).sum(dim=-1)
        if q_control is not None:
            c = c + ((action ** 2) * q_control.to(action)).sum(dim=-1)
        return c

    return dynamics, cost


# ---------------------------------------------------------------------------------------------
# Deterministic stand-in for STOCHASTIC dynamics (rollout_samples M > 1, mppi.py:334-373): the M
# rollouts of one action sequence differ by a fixed disturbance table w (M,T,nx) instead of by
# draws made inside the callback, so the live reference, the oracle and the engine's callback path
# see identical "randomness".  The callbacks get one batch of M*K rows, row = m*K + k (mppi.py:351);
# any other batch size (get_rollouts, M = 1) is treated as m = 0.  Step-dependent signature
# (state, u, t) -> construct the controller with step_dependent_dynamics=True.
# ---------------------------------------------------------------------------------------------
def make_linear_goal_multi(B, goal, w, K):
    M = w.shape[0]

    def dynamics(state, action, t):
        rows = state.shape[0]
        if rows == M * K:
            m = torch.arange(rows, device=state.device) // K
        else:
            m = torch.zeros(rows, dtype=torch.long, device=state.device)
        return state + action @ B.T + w[m, t]

    def cost(state, action, t):
        dx = goal - state
        return (dx ** 2).sum(dim=-1)

    def terminal(states, actions):
        dx = goal - states[..., -1, :]
        return (dx ** 2).sum(dim=-1)                    # (M,K) for (M,K,T,nx) states

    return dynamics, cost, terminal


# ---------------------------------------------------------------------------------------------
# Process noise with INJECTED draws: x' = f(x, u) + sd * w[m, k, t].  The callbacks of an M > 1
# command see one batch of M*K rows, row = m*K + k (mppi.py:351); w (M,K,T,nx) holds the draws the
# engine's fused kernel makes itself (oracle/philox.process_normals), so the oracle and the kernel
# see identical disturbances.  Step-dependent signature -> step_dependent_dynamics=True.
# ---------------------------------------------------------------------------------------------
def with_injected_process_noise(dynamics, w, sd):
    M, K = w.shape[0], w.shape[1]

    def noisy(state, action, t):
        assert state.shape[0] == M * K
        return dynamics(state, action) + sd * w[:, :, t].reshape(M * K, -1)

    return noisy
