"""pytorch_mppi_amd/trace.py: the operator vocabulary.  Each case is a dynamics / running-cost pair a user might write with
one family of torch operations; it must translate AND agree with the callable on random batches (host build, fp64, 1e-9):
activations of nn.Sequential networks, normalisations, losses, einsum / addmm / bmm, reductions with dims, cumulative and
reordering ops, the extra transcendental functions, logical masks."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from pytorch_mppi_amd import trace

NX, NU = 3, 2
Q = lambda s, a: (s ** 2).sum(-1)


def _seq(act):
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(NX + NU, 6), act, nn.Linear(6, NX)).double()
    return lambda s, a: s + net(torch.cat((s, a), 1))


ACTS = dict(relu=nn.ReLU(), relu_inplace=nn.ReLU(inplace=True), elu=nn.ELU(), celu=nn.CELU(0.7), gelu=nn.GELU(),
            gelu_tanh=nn.GELU(approximate="tanh"), selu=nn.SELU(), silu=nn.SiLU(), softplus=nn.Softplus(), sigmoid=nn.Sigmoid(),
            leaky=nn.LeakyReLU(0.1), hardtanh=nn.Hardtanh(-0.5, 0.8), relu6=nn.ReLU6(), tanhshrink=nn.Tanhshrink(),
            softsign=nn.Softsign(), mish=nn.Mish(), hardswish=nn.Hardswish(), hardsigmoid=nn.Hardsigmoid(),
            logsigmoid=nn.LogSigmoid(), softshrink=nn.Softshrink(0.3), hardshrink=nn.Hardshrink(0.3), threshold=nn.Threshold(0.1, -2.0))


@pytest.mark.parametrize("name", sorted(ACTS))
def test_activation_modules(name):
    f = _seq(ACTS[name])
    code = trace.generate(f, Q, NX, NU)
    assert trace.verify_on_host(code, f, Q, NX, NU)


def _cases():
    torch.manual_seed(1)
    W = torch.randn(NX, NX, dtype=torch.float64)
    ln = nn.LayerNorm(NX).double()
    with torch.no_grad():
        ln.weight.mul_(1.3)
        ln.bias.add_(0.2)
    u1 = lambda a: a[:, :1]
    c = {}
    c["layer_norm"] = (lambda s, a: s + 0.1 * ln(s) * u1(a), Q)
    c["softmax"] = (lambda s, a: s + F.softmax(s, dim=-1) * u1(a), Q)
    c["log_softmax_softmin"] = (lambda s, a: s + F.log_softmax(s, dim=1) * u1(a) + F.softmin(s, dim=1), Q)
    c["einsum_matrix"] = (lambda s, a: torch.einsum("bi,ij->bj", s, W) + u1(a), Q)
    c["einsum_quadratic_cost"] = (lambda s, a: s + u1(a), lambda s, a: torch.einsum("bi,ij,bj->b", s, W @ W.T, s))
    c["quadratic_cost_matmul"] = (lambda s, a: s + u1(a), lambda s, a: ((s @ (W @ W.T)) * s).sum(1))
    c["quadratic_cost_diag"] = (lambda s, a: s + u1(a), lambda s, a: (s @ (W @ W.T) @ s.T).diag())
    c["linalg_norm"] = (lambda s, a: s / (1.0 + torch.linalg.norm(s, dim=1, keepdim=True)) + u1(a),
                        lambda s, a: torch.linalg.vector_norm(s, ord=1, dim=1) + s.norm(p=3, dim=1) + torch.norm(a, dim=-1))
    c["normalize"] = (lambda s, a: F.normalize(s, dim=1) + u1(a), Q)
    c["cross"] = (lambda s, a: s + 0.1 * torch.cross(s, torch.cat((a, u1(a)), 1), dim=1) + 0.1 * torch.linalg.cross(s, s.roll(1, 1)), Q)

    def rot(s, a):
        co, sn = torch.cos(s[:, 2]), torch.sin(s[:, 2])
        R = torch.stack((torch.stack((co, -sn), -1), torch.stack((sn, co), -1)), -2)          # (B,2,2)
        v = torch.bmm(R, a.unsqueeze(-1)).squeeze(-1)
        return torch.cat((s[:, :2] + 0.1 * v, s[:, 2:] + 0.05 * u1(a)), 1)
    c["bmm_rotation"] = (rot, Q)
    c["square_sign_remainder_fmod"] = (lambda s, a: torch.square(s) * torch.sign(u1(a)) + torch.remainder(s, 2.0) + torch.fmod(u1(a), 0.7), Q)
    c["losses"] = (lambda s, a: s + u1(a),
                   lambda s, a: F.mse_loss(s, torch.zeros_like(s), reduction="none").sum(1) + F.l1_loss(a, torch.ones_like(a), reduction="none").sum(1)
                   + F.smooth_l1_loss(s, torch.zeros_like(s), reduction="none", beta=0.5).sum(1) + F.huber_loss(s, torch.ones_like(s), reduction="none", delta=0.7).sum(1))
    c["cumsum_cumprod"] = (lambda s, a: torch.cumsum(s, dim=1) + 0.1 * torch.cumprod(s, dim=1) + u1(a), Q)
    c["flip_roll"] = (lambda s, a: torch.flip(s, dims=(1,)) + torch.roll(s, 1, dims=1) * u1(a), Q)
    c["addmm_addcmul_lerp"] = (lambda s, a: torch.addmm(u1(a).expand(-1, NX), s, W, beta=0.5, alpha=2.0)
                               + torch.lerp(s, torch.ones_like(s), 0.3) + torch.addcmul(s, s, s, value=0.5) + torch.addcdiv(s, s, 2 + s ** 2), Q)
    c["expm1_log1p_log2_exp2"] = (lambda s, a: torch.expm1(0.1 * s) + torch.log1p(s ** 2) + torch.log2(1 + s ** 2) + torch.exp2(0.1 * s) + torch.log10(2 + s ** 2) + u1(a), Q)
    c["erf_atan_asin_acos"] = (lambda s, a: torch.erf(s) + torch.atan(s) + torch.asin(torch.tanh(s)) + torch.acos(torch.tanh(s)) + u1(a), Q)
    c["sinh_cosh_hypot_logaddexp"] = (lambda s, a: torch.sinh(0.1 * s) + torch.cosh(0.1 * s) + torch.hypot(s, u1(a)) + torch.logaddexp(s, u1(a)), Q)
    c["ceil_round_trunc_frac"] = (lambda s, a: torch.ceil(s) + torch.round(s * 3) + torch.trunc(s * 2) + torch.frac(s) + u1(a), Q)
    c["var_std_mean"] = (lambda s, a: s + s.var(dim=1, keepdim=True) + s.std(dim=1, keepdim=True, unbiased=False) * u1(a) + s.mean(1, keepdim=True), Q)
    c["max_min_with_dim"] = (lambda s, a: s + s.abs().max(dim=1, keepdim=True).values * u1(a) + torch.min(s, dim=1, keepdim=True)[0],
                             lambda s, a: s.max(dim=1).values ** 2 + a.amax(1))
    c["logical_masks"] = (lambda s, a: torch.where((s > 0) & (s < 1), s, -s) + torch.where((s < -1) | ~(s < 2), s * 0.5, s) + u1(a),
                          lambda s, a: torch.where(torch.logical_and(s[:, 0] > 0, s[:, 1] != 0.25), s[:, 0], s[:, 1] ** 2) + (s[:, 2] == s[:, 2]) * 1.0)
    outer = lambda s: s.unsqueeze(-1) * s.unsqueeze(-2)                       # per-sample (B, nx, nx)
    c["outer_tril_triu_diagonal"] = (lambda s, a: s + outer(s).tril().sum(1) * 0.1 + outer(s).triu(1).sum(2) * 0.2
                                     + outer(s).diagonal(dim1=-2, dim2=-1) * u1(a), Q)
    return c


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_operator_families(name):
    f, q = CASES[name]
    code = trace.generate(f, q, NX, NU)
    assert trace.verify_on_host(code, f, q, NX, NU)


def test_indices_of_max_are_refused():
    with pytest.raises(trace.TraceUnsupported):
        trace.generate(lambda s, a: s + s.max(dim=1, keepdim=True).indices * 1.0, Q, NX, NU)
    with pytest.raises(trace.TraceUnsupported):
        trace.generate(lambda s, a: s + torch.argmax(s, dim=1, keepdim=True) * 1.0, Q, NX, NU)


def test_user_idioms_fill_a_created_tensor_in_place_ops_and_masked_updates():
    """the ways people actually write dynamics: build the next state in a tensor they create, update columns in place, wrap
    an angle with a boolean mask (the reference's own angular_diff_batch, tests/pendulum_approximate.py:87-92)"""
    def f(s, a):
        nxt = torch.zeros(s.shape[0], NX, dtype=s.dtype, device=s.device)
        nxt[:, 0] = s[:, 0] + 0.1 * torch.cos(s[:, 2]) * a[:, 0]
        nxt[:, 1] = s[:, 1] + 0.1 * torch.sin(s[:, 2]) * a[:, 0]
        th = s[:, 2] + 0.3 * a[:, 1]
        th[th > math.pi] -= 2 * math.pi
        th[th < -math.pi] += 2 * math.pi
        nxt[:, 2] = th
        nxt[:, :2].mul_(0.99)
        nxt += torch.tensor([0.01, 0.0, 0.0], dtype=s.dtype, device=s.device)
        return nxt

    def q(s, a):
        c = (s[:, :2] ** 2).sum(1)
        c[s[:, 0].abs() > 1.0] = 50.0
        pen = a.clone().abs_().sum(1)
        pen.clamp_(max=1.5)
        return c + 0.1 * pen + s[:, 2].masked_fill(s[:, 2] < 0, 0.0)

    code = trace.generate(f, q, NX, NU)
    assert trace.verify_on_host(code, f, q, NX, NU)


def test_views_write_through_like_torch():
    def f(s, a):
        s = s.clone()
        v = s[:, 1:]                 # a view: the in-place update below changes s
        v += a
        return s

    code = trace.generate(f, Q, NX, NU)
    assert trace.verify_on_host(code, f, Q, NX, NU)


def test_random_draws_and_data_dependent_selections_are_refused():
    with pytest.raises(trace.TraceUnsupported, match="random"):
        trace.generate(lambda s, a: s + 0.1 * torch.randn_like(a[:, :1]), Q, NX, NU)
    with pytest.raises(trace.TraceUnsupported, match="random"):
        trace.generate(lambda s, a: s + 0.1 * torch.randn(s.shape[0], NX, dtype=s.dtype), Q, NX, NU)
    with pytest.raises(trace.TraceUnsupported, match="masked selection"):
        trace.generate(lambda s, a: s + s[s > 0].sum(), Q, NX, NU)


def test_saturating_activations_do_not_overflow_far_from_the_origin():
    """softplus / mish / logsigmoid are written through exp(): the traced forms must follow torch's own guards (softplus is
    linear above its threshold, logsigmoid is evaluated on -|x|), or states far from the origin would come back inf / nan --
    far beyond the range the random verification batches visit"""
    import ctypes as C
    import numpy as np

    def f(s, a):
        return torch.cat((F.softplus(s[:, :1] * 40.0, beta=2.0), F.mish(s[:, 1:2] * 50.0), F.logsigmoid(s[:, 2:3] * 400.0)), 1) + a[:, :1] * 0.0

    code = trace.generate(f, Q, NX, NU)
    assert trace.verify_on_host(code, f, Q, NX, NU)
    # evaluate the traced step far out through the same host build the verification uses
    x = torch.tensor([[30.0, 25.0, -20.0], [-30.0, -25.0, 20.0]], dtype=torch.float64)
    want = f(x, torch.zeros(2, NU, dtype=torch.float64))
    assert torch.isfinite(want).all()
    got = trace.evaluate_on_host(code, x.numpy(), np.zeros((2, NU)), NX, NU)[0]
    assert np.isfinite(got).all() and np.allclose(got, want.numpy(), rtol=1e-9, atol=1e-12)


def test_batch_norm_in_eval_mode_and_timestep_arithmetic():
    torch.manual_seed(4)
    bn = nn.BatchNorm1d(NX).double().eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.tensor([0.1, -0.2, 0.3]))
        bn.running_var.copy_(torch.tensor([1.5, 0.7, 2.0]))
        bn.weight.mul_(1.2)
    f = lambda s, a: s + 0.1 * bn(s) * a[:, :1]
    code = trace.generate(f, Q, NX, NU)
    assert trace.verify_on_host(code, f, Q, NX, NU)
    bn.train()
    with pytest.raises(trace.TraceUnsupported, match="training"):
        trace.generate(f, Q, NX, NU)
    # the timestep: t % period, t // k
    ft = lambda s, a, t: s + a[:, :1] * (1.0 + 0.1 * (t % 4)) + 0.01 * (t // 3)
    qt = lambda s, a, t: (s ** 2).sum(-1) * (1.0 + 0.05 * t)
    code = trace.generate(ft, qt, NX, NU, None, True)
    assert trace.verify_on_host(code, ft, qt, NX, NU, None, True)


def test_obstacle_costs_and_constructor_idioms():
    """distance-to-obstacle costs (cdist, any() of a mask), state.new_zeros(...), the `if not torch.is_tensor(x)` guard"""
    obstacles = torch.tensor([[0.5, 0.5], [-1.0, 0.3], [0.2, -0.8]], dtype=torch.float64)

    def f(s, a):
        if not torch.is_tensor(s):
            s = torch.tensor(s)
        nxt = s.new_zeros(s.shape[0], NX)
        nxt[:, :2] = s[:, :2] + 0.1 * a
        nxt[:, 2] = torch.fmax(s[:, 2], torch.mv(a, torch.tensor([0.3, -0.2], dtype=s.dtype)))
        return nxt

    def q(s, a):
        d = torch.cdist(s[:, None, :2], obstacles.to(s.device)).squeeze(1)               # (B, 3)
        hit = (d < 0.4).any(dim=1)
        return torch.exp(-d ** 2 / 0.1).sum(1) + 100.0 * hit + (d > 5.0).all(1) * 3.0 + torch.cdist(s[:, None, :2], obstacles, p=1.0).amin((1, 2))

    code = trace.generate(f, q, NX, NU)
    assert trace.verify_on_host(code, f, q, NX, NU)
    with pytest.raises(trace.TraceUnsupported, match="item"):
        trace.generate(lambda s, a: s * s[0, 0].item(), Q, NX, NU)
