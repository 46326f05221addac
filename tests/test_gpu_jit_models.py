"""GPU: user models compiled at run time (pytorch_mppi_amd/jit.py) -- the fused-path counterpart of
the reference's "any callable" plugin API (mppi.py:63-64)."""
import math

import numpy as np
import pytest
import torch

import jit_fixtures as jf
import pytorch_mppi_amd as pm

import golden_util as gu
import margins

pytestmark = pytest.mark.gpu


def test_jit_pendulum_is_bitwise_the_builtin_kernel():
    builtin, user = jf.pendulum_user()
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
    f, q, term = jf.unicycle_callables()
    model = jf.unicycle()
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
    r32 = orc.command(gu.problem_as(p, torch.float32), U0.float(), x0.float(), z.float(), True)     # the callables hold Python floats only
    tol = 1e-9 if dtype == torch.float64 else 1e-5
    for c in (fused, generic):
        c.inject_noise(z.to(dtype))
        a = c.command(x0.to(dtype).cuda())
        for name, got in (("action", a), ("U", c.U), ("cost_total", c.cost_total)):
            margins.check(f"jit unicycle {dtype} {'fused' if c is fused else 'generic'}", name, got.cpu().numpy(), r[name].numpy(),
                          r32[name].numpy() if dtype == torch.float32 else None, rtol=tol, scale_floor=1.0)
    assert fused.states.shape == (1, K, T, 3)
    assert torch.allclose(fused.states, generic.states, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_step_dependent_dynamics_run_fused(dtype):
    """step_dependent_dynamics=True (mppi.py:147-154) with a native model whose callables take t: the fused
    kernel (the device functor always sees the timestep), not the callback loop; against the callback path
    on the same callables and the fp64 oracle."""
    from oracle import mppi_oracle as orc
    f, q = jf.drifting_callables()
    model = jf.drifting()
    K, T = 900, 18
    g = torch.Generator().manual_seed(8)
    U0 = torch.randn(T, 2, generator=g, dtype=torch.float64) * 0.1
    x0 = torch.tensor([0.3, -0.2], dtype=torch.float64)
    sigma = torch.diag(torch.tensor([0.7, 1.2], dtype=torch.float64))
    z = torch.randn(K, T, 2, generator=g, dtype=torch.float64)
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=0.8, U_init=U0.to(dtype), step_dependent_dynamics=True)
    fused = pm.MPPI(model.dynamics, model.running_cost, 2, sigma.to(dtype), **kw)
    generic = pm.MPPI(f, q, 2, sigma.to(dtype), **kw)
    assert fused._model is model and not fused._needs_generic() and generic._needs_generic()
    p = orc.Problem(dynamics=f, running_cost=q, nx=2, noise_sigma=sigma, K=K, T=T, lambda_=0.8, step_dependent_dynamics=True)
    r = orc.command(p, U0, x0, z, True)
    r32 = orc.command(gu.problem_as(p, torch.float32), U0.float(), x0.float(), z.float(), True)
    tol = 1e-9 if dtype == torch.float64 else 1e-5
    for c in (fused, generic):
        c.inject_noise(z.to(dtype))
        a = c.command(x0.to(dtype).cuda())
        for name, got in (("action", a), ("U", c.U), ("cost_total", c.cost_total)):
            margins.check(f"jit step-dependent {dtype} {'fused' if c is fused else 'generic'}", name, got.cpu().numpy(), r[name].numpy(),
                          r32[name].numpy() if dtype == torch.float32 else None, rtol=tol, scale_floor=1.0)
    # a time-independent native model keeps refusing the flag's mismatch: callback path
    m2 = pm.models.Integrator(2, 2)
    c2 = pm.MPPI(lambda s, a, t: m2.dynamics(s, a), lambda s, a, t: m2.running_cost(s, a), 2, sigma.to(dtype), **kw)
    assert c2._needs_generic()


def test_jit_model_under_kmppi_interpolates_inside_k1():
    """A user model with four controls under KMPPI: its own shared object carries the KMPPI-fused K1
    (csrc/rollout_kmppi.hpp, reached through mppi_register_model's launcher); checked against the generic
    callback path and the fp64 oracle on the same draw."""
    from oracle import mppi_oracle as orc
    f, q = jf.cart4_callables()
    model = jf.cart4()
    K, T, S, nu, nx = 1500, 28, 14, 4, 6
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(nx, generator=g, dtype=torch.float64) * 0.3
    sigma = torch.diag(torch.rand(nu, generator=g, dtype=torch.float64) + 0.4)
    umax = torch.full((nu,), 1.5, dtype=torch.float64)
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=2.0, u_max=umax.float(), num_support_pts=S,
              U_init=torch.zeros(T, nu))
    fused = pm.KMPPI(model.dynamics, model.running_cost, nx, sigma.float(), **kw)
    generic = pm.KMPPI(f, q, nx, sigma.float(), **kw)
    assert not fused._needs_generic() and generic._needs_generic() and fused._fused_interp_expected()
    p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sigma, K=K, T=T, lambda_=2.0, u_max=umax)
    W, W_shift, _, _ = orc.kmppi_matrices(T, S, torch.float64)
    theta, U = torch.zeros(S, nu, dtype=torch.float64), torch.zeros(T, nu, dtype=torch.float64)
    lib = pm._native.lib()
    for s in range(2):
        z = torch.randn(K, S, nu, generator=g, dtype=torch.float64)
        r = orc.kmppi_command(p, theta, U, x0, z, W, W_shift, True)
        r32 = orc.kmppi_command(gu.problem_as(p, torch.float32), theta.float(), U.float(), x0.float(), z.float(), W.float(),
                                W_shift.float(), True)
        theta, U = r["theta"], r["U"]
        n0 = lib.mppi_stat_kmppi_fused_rollouts()
        for c in (fused, generic):
            c.inject_noise(z.float())
            a = c.command(x0.float().cuda())
            for name, got in (("action", a), ("U", c.U), ("theta", c.theta), ("cost_total", c.cost_total)):
                margins.check(f"jit cart4 under KMPPI {'fused' if c is fused else 'generic'}", f"step {s} {name}", got.cpu().numpy(),
                              r[name].numpy(), r32[name].numpy(), rtol=1e-5, scale_floor=1.0)
        assert lib.mppi_stat_kmppi_fused_rollouts() == n0 + 1       # the fused controller took the in-kernel interpolation
