"""KMPPI: the sequences of the NEXT command's shift made beside U = W theta in the update's last launch (C-ABI 22
mppi_kmppi_after_update; reference mppi.py:617-619, :232-238, :682) -- the bits of the explicit shift, and dropped whenever
U / theta / u_init / the horizon are not what that launch saw."""
import pytest
import torch

import pytorch_mppi_amd as pm

pytestmark = pytest.mark.gpu


def _mk(monkeypatch, ahead, dtype=torch.float32, **kw):
    monkeypatch.setenv("MPPI_KMPPI_SHIFT_AHEAD", "1" if ahead else "0")
    nx, nu, K, T = 8, 4, 4096, 20
    m = pm.models.Integrator(nx, nu)
    torch.manual_seed(3)
    return pm.KMPPI(m.dynamics, m.running_cost, nx, (torch.eye(nu) * 0.4).to(dtype), num_samples=K, horizon=T, device="cuda", lambda_=20.0,
                    num_support_pts=7, kernel=pm.RBFKernel(sigma=1.5), rng="philox", seed=5, u_init=torch.full((nu,), 0.3, dtype=dtype), **kw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_shift_ahead_commands_the_bits_of_the_explicit_shift(monkeypatch, dtype):
    a, b = _mk(monkeypatch, True, dtype), _mk(monkeypatch, False, dtype)
    assert a.shift_ahead and not b.shift_ahead
    x = torch.linspace(-1, 1, 8, device="cuda", dtype=dtype)
    for i in range(6):
        shift = i != 3
        assert torch.equal(a.command(x, shift_nominal_trajectory=shift), b.command(x, shift_nominal_trajectory=shift)), i
        assert torch.equal(a.U, b.U) and torch.equal(a.theta, b.theta)
        assert a._shift_ready is not None and b._shift_ready is None


def test_shift_ahead_is_dropped_when_its_inputs_move(monkeypatch):
    a, b = _mk(monkeypatch, True), _mk(monkeypatch, False)
    x = torch.zeros(8, device="cuda")

    def both(f):
        for c in (a, b):
            torch.manual_seed(9)          # (reset() draws the nominal sequence from torch's generator)
            f(c)
        assert torch.equal(a.command(x), b.command(x)) and torch.equal(a.theta, b.theta)
    both(lambda c: None)
    both(lambda c: c.reset())                                                           # theta zeroed IN PLACE, U re-drawn
    both(lambda c: setattr(c, "u_init", torch.full((4,), -0.7, device="cuda")))        # another u_init
    both(lambda c: c.u_init.mul_(0.5))                                                  # ... and written in place
    both(lambda c: setattr(c, "theta", c.theta * 0.5))                                  # the caller's own control points
    both(lambda c: c.theta.add_(0.1))
    both(lambda c: c.change_horizon(24))
    both(lambda c: c.shift_nominal_trajectory())                                        # an explicit shift between two commands
