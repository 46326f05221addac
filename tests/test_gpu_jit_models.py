"""GPU: user models compiled at run time (pytorch_mppi_amd/jit.py) -- the fused-path counterpart of
the reference's "any callable" plugin API (mppi.py:63-64)."""
import math

import numpy as np
import pytest
import torch

import pytorch_mppi_amd as pm
from pytorch_mppi_amd import jit

pytestmark = pytest.mark.gpu


def test_jit_pendulum_is_bitwise_the_builtin_kernel():
    builtin = pm.models.Pendulum()
    user = jit.compile_model(
        "pendulum_user", 2, 1, dynamics=builtin.dynamics, running_cost=builtin.running_cost,
        step="const T uc = clampT(u[0], T(-2), T(2));"
             "T nthd = x[1] + (T(15) * m_sin(x[0]) + T(3) * uc) * T(0.05);"
             "nthd = clampT(nthd, T(-8), T(8)); x[0] = x[0] + nthd * T(0.05); x[1] = nthd;",
        cost="const T pi = T(3.141592653589793), two_pi = T(6.283185307179586);"
             "T r = m_fmod(x[0] + pi, two_pi); if (r != T(0) && r < T(0)) r += two_pi;"
             "const T an = r - pi; return an * an + T(0.1) * (x[1] * x[1]);")
    z = torch.randn(1000, 20, 1, generator=torch.Generator().manual_seed(0))
    outs = []
    for m in (builtin, user):
        c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(10.0), num_samples=1000, horizon=20, device="cuda",
                    u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), U_init=torch.zeros(20, 1))
        assert c._model is m and not c._needs_generic()
        c.inject_noise(z)
        outs.append((c.command(torch.tensor([math.pi, 1.0])), c.cost_total))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_jit_unicycle_fused_equals_callback_path_and_oracle(dtype):
    """a model the engine has never seen: unicycle with parameters + terminal cost"""
    from oracle import mppi_oracle as orc
    dt_, gx, gy, wT = 0.1, 1.5, -0.5, 3.0

    def f(s, a):
        return torch.stack((s[:, 0] + dt_ * a[:, 0] * torch.cos(s[:, 2]), s[:, 1] + dt_ * a[:, 0] * torch.sin(s[:, 2]),
                            s[:, 2] + dt_ * a[:, 1]), dim=1)

    def q(s, a):
        return (s[:, 0] - gx) ** 2 + (s[:, 1] - gy) ** 2 + 0.01 * (a ** 2).sum(-1)

    def term(states, actions):
        last = states[..., -1, :]
        return wT * ((last[..., 0] - gx) ** 2 + (last[..., 1] - gy) ** 2)

    model = jit.compile_model(
        "unicycle", 3, 2, dynamics=f, running_cost=q, terminal_state_cost=term, params=[dt_, gx, gy, wT],
        step="const T c = m_cos(x[2]), s = m_sin(x[2]); x[0] += p[0] * u[0] * c; x[1] += p[0] * u[0] * s; x[2] += p[0] * u[1];",
        cost="const T dx = x[0] - p[1], dy = x[1] - p[2]; return dx * dx + dy * dy + T(0.01) * (u[0] * u[0] + u[1] * u[1]);",
        terminal="const T dx = x[0] - p[1], dy = x[1] - p[2]; return p[3] * (dx * dx + dy * dy);")
    K, T = 777, 25
    g = torch.Generator().manual_seed(3)
    U0 = torch.randn(T, 2, generator=g, dtype=torch.float64) * 0.1
    x0 = torch.tensor([0.0, 0.0, 0.3], dtype=torch.float64)
    sigma = torch.diag(torch.tensor([0.5, 1.0], dtype=torch.float64))
    umax = torch.tensor([1.0, 2.0], dtype=torch.float64)
    z = torch.randn(K, T, 2, generator=g, dtype=torch.float64)
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=0.5, u_max=umax.to(dtype), U_init=U0.to(dtype))
    fused = pm.MPPI(model.dynamics, model.running_cost, 3, sigma.to(dtype), terminal_state_cost=model.terminal_state_cost, **kw)
    generic = pm.MPPI(f, q, 3, sigma.to(dtype), terminal_state_cost=term, **kw)
    assert not fused._needs_generic() and generic._needs_generic()
    p = orc.Problem(dynamics=f, running_cost=q, nx=3, noise_sigma=sigma, K=K, T=T, lambda_=0.5, u_max=umax, terminal_state_cost=term)
    r = orc.command(p, U0, x0, z, True)
    tol = 1e-9 if dtype == torch.float64 else 2e-5
    for c in (fused, generic):
        c.inject_noise(z.to(dtype))
        a = c.command(x0.to(dtype).cuda())
        for name, got in (("action", a), ("U", c.U), ("cost_total", c.cost_total)):
            ref = r[name].numpy()
            np.testing.assert_allclose(got.cpu().double().numpy(), ref, rtol=tol, atol=tol * max(1.0, np.abs(ref).max()), err_msg=name)
    assert fused.states.shape == (1, K, T, 3)
    assert torch.allclose(fused.states, generic.states, rtol=tol, atol=tol)
