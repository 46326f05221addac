"""M > 1 rollouts per action sequence (reference mppi.py:334-373) with ONE WAVE PER ROLLOUT COPY (csrc/rollout_copies.hpp, round 6)
against the one-lane-holds-all-copies form it replaces where it applies (rollout_stream_multi; MPPI_MULTI_COPIES=0): the same
process-noise stream, the same order of every sum -- bit for bit in fp32, to 1e-9 in fp64 -- for MPPI / SMPPI / KMPPI's two-launch form, M = 2..4, ragged K,
bounds, the null-action row, a terminal cost, per-sample states, fp32 and fp64.  (Both forms against the fp64 oracle fed the exported
process normals: tests/test_gpu_parity.py's multi-rollout tests run whichever form the engine picks -- this one.)"""
import pytest
import torch

import pytorch_mppi_amd as pm

pytestmark = pytest.mark.gpu


def _mk(cls, M, K, T, nx, nu, dtype, rng, **kw):
    m = pm.models.Integrator(nx, nu).with_process_noise(0.05)
    torch.manual_seed(4)
    extra = {}
    if cls is pm.KMPPI:
        extra.update(num_support_pts=6)
    if cls is pm.SMPPI:
        extra.update(action_min=-torch.ones(nu, dtype=dtype), action_max=torch.ones(nu, dtype=dtype), w_action_seq_cost=0.7, delta_t=0.2)
    return cls(m.dynamics, m.running_cost, nx, (torch.eye(nu) * 0.6).to(dtype), num_samples=K, horizon=T, device="cuda", lambda_=25.0,
               rollout_samples=M, rollout_var_cost=0.3, rollout_var_discount=0.9, rng=rng, seed=77, **extra, **kw)


@pytest.mark.parametrize("cls", [pm.MPPI, pm.SMPPI, pm.KMPPI])
@pytest.mark.parametrize("M,K,T,dtype,rng", [(3, 20000, 24, torch.float32, "philox"), (2, 4097, 9, torch.float32, "torch"),
                                             (4, 8192, 33, torch.float32, "philox"), (3, 3000, 12, torch.float64, "philox")])
def test_one_wave_per_copy_commands_the_bits_of_the_one_lane_form(monkeypatch, cls, M, K, T, dtype, rng):
    nx, nu = 8, 4
    kw = dict(u_min=-torch.ones(nu, dtype=dtype) * 0.9, u_max=torch.ones(nu, dtype=dtype) * 1.1) if cls is not pm.SMPPI else {}
    if cls is pm.MPPI:
        kw.update(sample_null_action=True)
    outs = []
    for knob in ("1", "0"):
        monkeypatch.setenv("MPPI_MULTI_COPIES", knob)
        torch.manual_seed(9)
        c = _mk(cls, M, K, T, nx, nu, dtype, rng, **kw)
        assert not c._needs_generic()
        x = torch.linspace(-1, 1, nx, device="cuda", dtype=dtype)
        acts = []
        for i in range(3):
            acts.append(c.command(x, shift_nominal_trajectory=i != 1).clone())
        outs.append((torch.stack(acts), c.U.clone(), c.cost_total.clone(), c.omega.clone()))
    for u, v in zip(*outs):
        if dtype == torch.float32:
            assert torch.equal(u, v)
        else:
            # (fp64: the two kernels' multiply-adds are contracted differently in two places -- 1 ulp, amplified by the softmax)
            assert float((u - v).abs().max()) <= 1e-9 * max(1.0, float(v.abs().max()))


def test_per_sample_states_and_what_keeps_the_one_lane_form(monkeypatch):
    """per-sample initial states go through the copies kernel; a `states` read and sampler rows do not (their conditional stores /
    loads stay with rollout_stream_multi): same results either way"""
    M, K, T, nx, nu = 3, 6000, 10, 8, 4
    X = torch.randn(K, nx, generator=torch.Generator().manual_seed(2)).cuda()
    res = []
    for knob in ("1", "0"):
        monkeypatch.setenv("MPPI_MULTI_COPIES", knob)
        c = _mk(pm.MPPI, M, K, T, nx, nu, torch.float32, "philox")
        a = c.command(X).clone()
        res.append((a, c.cost_total.clone(), c.states.clone()))
    assert all(torch.equal(u, v) for u, v in zip(*res)) and res[0][2].shape == (M, K, T, nx)
