"""The on-chip rng="philox" command (csrc/rollout_onchip.hpp, include/mppi_amd.h ABI 18): no (K,T,nu) array -- one launch
generates, rolls out, keeps the bounded noise in accumulation registers / LDS and leaves a partial record per workgroup,
a second one combines them.  Checked against
  * the streaming form of the SAME command (generator launch -> K1 -> K3 -> K4; same seed => same Philox stream => the
    same normals): cost_total, U, action and omega to 1e-5 of their scale;
  * the fp64 oracle on the normals of that stream (device_philox_normals), SURVEY 7.3 criterion, margins to the ledger;
over shapes that exercise every storage class of the kernel (accumulation registers only / + LDS / + second generation),
ragged K, short and long horizons, control widths with 1..5 rows per super-step, bounds, the null-action row,
per-sample initial states, terminal cost, |noise| cost, u_per_command > 1, shift on/off, several commands in a row."""
import numpy as np
import pytest
import torch

import gpu_util
import margins

pytestmark = pytest.mark.gpu


def _models(kind, nx, nu):
    import pytorch_mppi_amd as pm
    from oracle import dynamics as dyn
    if kind == "pendulum":
        return pm.models.Pendulum(), (lambda dt: (dyn.pendulum_dynamics, dyn.pendulum_cost))
    if kind == "integrator":
        return pm.models.Integrator(nx, nu), (lambda dt: dyn.make_quadtoy(nx, nu))
    if kind == "linear":
        g = torch.Generator().manual_seed(11)
        B, goal = torch.randn(nx, nu, generator=g) * 0.3, torch.randn(nx, generator=g)
        return pm.models.LinearGoal(B, goal), (lambda dt: dyn.make_linear_goal(B.to(dt), goal.to(dt))[:2])
    raise ValueError(kind)


def _make(kind, nx, nu, K, T, onchip, lam=1.0, seed=99, sigma=None, **kw):
    import pytorch_mppi_amd as pm
    model, mk = _models(kind, nx, nu)
    seed += margins.seed_offset()
    g = torch.Generator().manual_seed(5 + margins.seed_offset())
    U0 = torch.randn(T, nu, generator=g) * 0.1
    sigma = sigma if sigma is not None else (torch.eye(nu) * 0.7 if nu > 1 else torch.tensor(0.7))
    c = pm.MPPI(model.dynamics, model.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", lambda_=lam,
                U_init=U0.clone(), rng="philox", seed=seed, **kw)
    c.philox_onchip = onchip
    return c, mk, sigma, U0


def _onchip_count():
    from pytorch_mppi_amd import _native as N
    return int(N.lib().mppi_stat_onchip_commands())


CASES = [
    # kind, nx, nu, K, T, extra ctor kwargs      (built-in fused models: see csrc/rollout_*.hip for the (nx, nu) lists)
    ("integrator", 16, 12, 65536, 64, {}),                                   # C3: registers + LDS + second generation
    ("integrator", 16, 12, 1000, 64, {}),                                    # ragged K (tail workgroup 232 of 256 lanes)
    ("integrator", 16, 12, 20000, 7, {}),                                    # short horizon: registers only, one partial tile
    ("integrator", 6, 4, 30000, 100, dict(u_min=torch.tensor([-0.5] * 4), u_max=torch.tensor([0.8] * 4))),   # 1 row per step, LDS
    ("integrator", 8, 4, 17000, 50, dict(sample_null_action=True)),          # the null-action row
    ("integrator", 12, 6, 17000, 40, dict(noise_abs_cost=True, u_per_command=3)),   # nu = 6: 3 rows = 2 timesteps per super-step
    ("integrator", 4, 2, 40000, 130, {}),                                    # nu = 2: two timesteps per row
    ("linear", 10, 3, 17000, 33, dict(u_scale=2.0)),                         # nu = 3: 3 rows = 4 timesteps per super-step
    ("linear", 6, 3, 17000, 21, dict(sample_null_action=True, u_min=torch.tensor([-1.0] * 3), u_max=torch.tensor([0.7] * 3))),
    ("pendulum", 2, 1, 20000, 48, dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0))),   # nu = 1: four timesteps per row
    ("integrator", 4, 2, 300, 300, {}),                                      # long horizon at small K (forced on-chip); the pendulum is
                                                                             # chaotic over 300 steps: its fp32 floor swallows any bound
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-nx{c[1]}-nu{c[2]}-K{c[3]}-T{c[4]}" for c in CASES])
def test_onchip_command_matches_streaming_command_and_fp64_oracle(case):
    from oracle import mppi_oracle as orc
    kind, nx, nu, K, T, kw = case
    kw = dict(kw)
    sig = kw.pop("sigma", None)
    # a healthy lambda from a probe command (cost spread of this problem)
    probe, _, _, _ = _make(kind, nx, nu, K, T, False, sigma=sig, **kw)
    x0 = torch.randn(nx, generator=torch.Generator().manual_seed(3 + margins.seed_offset()))
    probe.command(x0.cuda())
    lam = float(probe.cost_total.double().std()) + 1e-3
    del probe
    a, mk, sigma, U0 = _make(kind, nx, nu, K, T, True, lam=lam, sigma=sig, **kw)
    b, _, _, _ = _make(kind, nx, nu, K, T, False, lam=lam, sigma=sig, **kw)
    n0 = _onchip_count()
    for step, shift in enumerate((True, False, True)):
        U_before = a.U.clone()
        act_a = a.command(x0.cuda(), shift_nominal_trajectory=shift)
        act_b = b.command(x0.cuda(), shift_nominal_trajectory=shift)
        assert a.last_draw == "philox-onchip", a.last_draw
        assert b.last_draw in ("philox-fill", "philox-k1"), b.last_draw
        # fp64 / fp32 oracle on the normals of the stream
        z = gpu_util.device_philox_normals(a, a._call)
        out = []
        for dt in (torch.float64, torch.float32):
            f, q = mk(dt)
            cast = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
            p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sigma.to(dt), K=K, T=T, lambda_=lam, **cast)
            out.append(orc.command(p, U_before.cpu().to(dt), x0.to(dt), z.to(dt), shift))
        r64, r32 = out
        got = dict(action=act_a, U=a.U, cost_total=a.cost_total, omega=a.omega)
        for key in got:
            margins.check(f"onchip/{kind}-nu{nu}-K{K}-T{T}/step{step}", key, got[key].detach().cpu().numpy(),
                          r64[key].numpy(), r32[key].numpy(), rtol=1e-5)
        # against the streaming command: two differently scheduled kernels (fma contraction, order of the cost terms), so
        # the bound is the parity rule itself -- 1e-5 of the scale, or twice the reference's own fp32 floor where that is
        # larger (the pendulum's wrapped angle amplifies a 1-ulp difference of an action over the horizon)
        for name, xa, xb in (("cost_total", a.cost_total, b.cost_total), ("U", a.U, b.U), ("action", act_a, act_b), ("omega", a.omega, b.omega)):
            s = max(float(xb.abs().max()), 1e-30)
            floor = float((r32[name].double() - r64[name]).abs().max())
            assert float((xa - xb).abs().max()) <= max(1e-5 * s, 2 * floor), (name, step)
        # the lazily materialised arrays come from the same stream
        if step == 0 and K * T * nu <= 4_000_000:
            assert torch.allclose(a.noise, b.noise, rtol=0, atol=1e-6)
            assert torch.allclose(a.perturbed_action, b.perturbed_action, rtol=0, atol=1e-6)
    assert _onchip_count() - n0 == 3, "every command of the first controller ran in the on-chip form"


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-nx{c[1]}-nu{c[2]}-K{c[3]}-T{c[4]}" for c in CASES])
def test_rows_waiting_in_memory_change_no_bit(case):
    """ABI 20: the rows that fit neither registers nor LDS WAIT in `onchip_spill` (stored behind the rollout, fetched in the
    weighting phase) instead of being generated a second time.  The same values enter the same column sums in the same order:
    every output is bit for bit what the twice-generating form gives, over every storage class of the kernel, bounds, the
    null-action row, short horizons (nothing to spill: no array), shift on / off."""
    from pytorch_mppi_amd import _native as N
    kind, nx, nu, K, T, kw = case
    kw = dict(kw)
    sig = kw.pop("sigma", None)
    a, _, _, _ = _make(kind, nx, nu, K, T, True, lam=25.0, sigma=sig, **kw)
    b, _, _, _ = _make(kind, nx, nu, K, T, True, lam=25.0, sigma=sig, **kw)
    a.onchip_spill, b.onchip_spill = True, False
    x0 = torch.randn(nx, generator=torch.Generator().manual_seed(3)).cuda()
    for shift in (True, False, True):
        ua, ub = a.command(x0, shift_nominal_trajectory=shift), b.command(x0, shift_nominal_trajectory=shift)
        assert a.last_draw == b.last_draw == "philox-onchip"
        assert int(N.lib().mppi_last_command_form()) == N.FORM_ONCHIP
        for name, xa, xb in (("action", ua, ub), ("U", a.U, b.U), ("cost_total", a.cost_total, b.cost_total), ("omega", a.omega, b.omega)):
            assert torch.equal(xa, xb), (name, shift, float((xa - xb).abs().max()))
    want = int(N.lib().mppi_onchip_spill_elems(a._last))
    assert (a._spill[1] is None) == (want == 0) and b._spill is None
    if want:
        # an array that holds only part of the rows: those wait in memory, the rest is generated a second time, memory tiles
        # consumed between the regenerated ones -- still the same bits
        c, _, _, _ = _make(kind, nx, nu, K, T, True, lam=25.0, sigma=sig, **kw)
        c._spill = ((c.K_local, T, nu), torch.empty(max(want * 2 // 5, 4), device="cuda"))
        d, _, _, _ = _make(kind, nx, nu, K, T, True, lam=25.0, sigma=sig, **kw)
        d.onchip_spill = False
        for shift in (True, False):
            uc, ud = c.command(x0, shift_nominal_trajectory=shift), d.command(x0, shift_nominal_trajectory=shift)
            assert int(c._last.onchip_spill_elems) == c._spill[1].numel() < want
            for name, xa, xb in (("action", uc, ud), ("U", c.U, d.U), ("cost_total", c.cost_total, d.cost_total)):
                assert torch.equal(xa, xb), ("partial array", name, shift)
    if want:
        assert a._spill[1].numel() == want and int(a._last.onchip_spill) == a._spill[1].data_ptr()


PAIR_CASES = [
    # K, T, lambda, extra ctor kwargs -- all on the model the two-wave kernel is instantiated for (integrator 16 x 12)
    (65536, 64, None, {}),                                                                                   # C3, healthy softmax
    (65536, 64, 0.05, {}),                                                                                   # peaked: dead waves skip their tiles
    (50000, 64, None, dict(sample_null_action=True, u_min=torch.tensor([-0.4] * 12), u_max=torch.tensor([0.6] * 12))),   # ragged K, bounds, null row
    (65536, 48, None, dict(sample_null_action=True)),                                                        # 12 chunks: one regenerated tile
    (49152, 33, None, dict(u_min=torch.tensor([-0.5] * 12), u_max=torch.tensor([0.5] * 12))),                # odd chunk count: the odd wave owns one fewer
    (49152, 100, None, {}),                                                                                  # 25 chunks, three regenerated tiles
    (3000, 70, None, dict(u_per_command=2)),                                                                 # small K (forced on chip), partial last chunk
    (65536, 64, None, dict(noise_abs_cost=True, u_scale=1.5, sample_null_action=True)),                      # the terms of the other instantiation:
    (49152, 50, None, dict(u_scale=0.5, u_min=torch.tensor([-0.5] * 12), u_max=torch.tensor([0.5] * 12))),   #   chunks of three super-steps
]


@pytest.mark.parametrize("case", PAIR_CASES, ids=[f"K{c[0]}-T{c[1]}-lam{c[2]}-{'-'.join(c[3]) or 'plain'}" for c in PAIR_CASES])
def test_two_waves_per_sample_group_change_no_bit(case, monkeypatch):
    """Round 6 (csrc/rollout_onchip_pair.hpp): the on-chip K1 with two waves per 64-sample group -- alternating chunks of the horizon,
    each wave generating, keeping and summing its own rows, the rollout's state handed from one to the other through LDS.  Same
    Philox counters, same arithmetic per row, every sum in the one-wave kernel's order: every output bit for bit, over chunk counts
    even and odd, rows regenerated in the weighting phase (0..3 tiles), bounds, the null-action row, ragged K, a peaked softmax,
    shift on / off, several commands in a row -- and the counter says which kernel ran."""
    from pytorch_mppi_amd import _native as N
    K, T, lam, kw = case
    lib = N.lib()
    x0 = torch.randn(16, generator=torch.Generator().manual_seed(3)).cuda() * 0.3
    if lam is None:
        probe, _, _, _ = _make("integrator", 16, 12, K, T, True, lam=1.0, **kw)
        probe.command(x0)
        c = probe.cost_total
        lam = max(float((c - c.min()).median()), 1e-3)                   # a healthy softmax for this problem's cost spread
    a, _, _, _ = _make("integrator", 16, 12, K, T, True, lam=lam, **kw)
    b, _, _, _ = _make("integrator", 16, 12, K, T, True, lam=lam, **kw)
    for step, shift in enumerate((True, False, True)):
        monkeypatch.setenv("MPPI_ONCHIP_PAIR", "1")
        n0 = int(lib.mppi_stat_onchip_pair_launches())
        ua = a.command(x0, shift_nominal_trajectory=shift)
        torch.cuda.synchronize()
        assert int(lib.mppi_stat_onchip_pair_launches()) == n0 + 1 and a.last_draw == "philox-onchip", "the two-wave kernel took it"
        monkeypatch.setenv("MPPI_ONCHIP_PAIR", "0")
        ub = b.command(x0, shift_nominal_trajectory=shift)
        torch.cuda.synchronize()
        assert int(lib.mppi_stat_onchip_pair_launches()) == n0 + 1 and int(lib.mppi_last_command_form()) == N.FORM_ONCHIP
        for name, xa, xb in (("action", ua, ub), ("U", a.U, b.U), ("cost_total", a.cost_total, b.cost_total), ("omega", a.omega, b.omega)):
            assert torch.equal(xa, xb), (name, step, float((xa - xb).abs().max()))
        if step == 0:
            n_eff = float(1.0 / (a.omega.double() ** 2).sum())
            assert (n_eff > 8) == (case[2] is None), n_eff                 # the healthy cases are healthy, the peaked one is peaked


def test_smppi_on_the_two_wave_kernel_changes_no_bit(monkeypatch):
    """SMPPI (mppi.py:451-570) at C3's shape: base sequence, 1/dt rescaling of the bounded noise, the smoothness cost -- whose operand,
    the previous timestep's action, rides in the hand-over between the waves of a pair."""
    import pytorch_mppi_amd as pm
    from pytorch_mppi_amd import _native as N
    lib = N.lib()
    model = pm.models.Integrator(16, 12)
    x0 = torch.randn(16, generator=torch.Generator().manual_seed(4)).cuda() * 0.3

    def make():
        c = pm.SMPPI(model.dynamics, model.running_cost, 16, torch.eye(12) * 0.5, num_samples=65536, horizon=64, device="cuda", lambda_=2000.0,
                     rng="philox", seed=21 + margins.seed_offset(), w_action_seq_cost=3.0, delta_t=0.5,
                     action_min=torch.tensor([-1.0] * 12), action_max=torch.tensor([1.0] * 12))
        c.philox_onchip = True
        return c
    a, b = make(), make()
    for step in range(3):
        monkeypatch.setenv("MPPI_ONCHIP_PAIR", "1")
        n0 = int(lib.mppi_stat_onchip_pair_launches())
        ua = a.command(x0)
        torch.cuda.synchronize()
        assert int(lib.mppi_stat_onchip_pair_launches()) == n0 + 1 and a.last_draw == "philox-onchip"
        monkeypatch.setenv("MPPI_ONCHIP_PAIR", "0")
        ub = b.command(x0)
        torch.cuda.synchronize()
        assert int(lib.mppi_stat_onchip_pair_launches()) == n0 + 1 and b.last_draw == "philox-onchip"
        for name, xa, xb in (("action", ua, ub), ("U", a.U, b.U), ("action_sequence", a.action_sequence, b.action_sequence),
                             ("cost_total", a.cost_total, b.cost_total), ("omega", a.omega, b.omega)):
            assert torch.equal(xa, xb), (name, step, float((xa - xb).abs().max()))


def test_what_the_two_wave_kernel_does_not_take_stays_on_the_one_wave_kernel(monkeypatch):
    """no spill array, another model: the one-wave kernel (still on chip)."""
    from pytorch_mppi_amd import _native as N
    lib = N.lib()
    monkeypatch.setenv("MPPI_ONCHIP_PAIR", "1")
    c, _, _, _ = _make("integrator", 16, 12, 49152, 64, True, lam=30.0)
    c.onchip_spill = False
    n0, m0 = int(lib.mppi_stat_onchip_pair_launches()), _onchip_count()
    c.command(torch.zeros(16).cuda())
    assert int(lib.mppi_stat_onchip_pair_launches()) == n0 and _onchip_count() == m0 + 1
    c, _, _, _ = _make("integrator", 12, 6, 49152, 64, True, lam=30.0)
    n0, m0 = int(lib.mppi_stat_onchip_pair_launches()), _onchip_count()
    c.command(torch.zeros(12).cuda())
    assert int(lib.mppi_stat_onchip_pair_launches()) == n0 and _onchip_count() == m0 + 1


def test_spill_array_size_at_c3():
    """87 of a sample's 192 rows-of-4 wait in memory at C3: 25 super-steps in registers, 10 in LDS, the other 29 -- as 6 whole
    weighting tiles = 30 super-steps of 3 rows each, the horizon's last, partial tile included -- in the array: 90 rows x 65536
    samples x 16 B."""
    import ctypes as C
    from pytorch_mppi_amd import _native as N
    p = N.MppiProblem()
    p.K, p.T, p.nx, p.nu, p.dtype, p.sigma_diagonal = 65536, 64, 16, 12, N.F32, 1
    assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == 90 * 65536 * 4
    p.T = 20                                       # 20 super-steps: everything fits the registers
    assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == 0
    p.T, p.dtype = 64, N.F64
    assert int(N.lib().mppi_onchip_spill_elems(C.byref(p))) == 0


def test_onchip_per_sample_states_and_terminal_cost():
    import pytorch_mppi_amd as pm
    nx, nu, K, T = 12, 4, 20000, 30
    g = torch.Generator().manual_seed(1)
    model = pm.models.LinearGoal(torch.randn(nx, nu, generator=g) * 0.3, torch.randn(nx, generator=g))
    U0 = torch.randn(T, nu, generator=g) * 0.1
    X0 = torch.randn(K, nx, generator=g)

    def make(onchip):
        c = pm.MPPI(model.dynamics, model.running_cost, nx, torch.eye(nu) * 0.5, num_samples=K, horizon=T, device="cuda",
                    lambda_=20.0, U_init=U0.clone(), rng="philox", seed=7, terminal_state_cost=model.terminal_state_cost)
        c.philox_onchip = onchip
        return c
    a, b = make(True), make(False)
    n0 = _onchip_count()
    aa, ab = a.command(X0.cuda()), b.command(X0.cuda())
    assert _onchip_count() - n0 == 1 and a.last_draw == "philox-onchip"
    assert float((a.cost_total - b.cost_total).abs().max()) <= 1e-5 * float(b.cost_total.abs().max())
    assert float((aa - ab).abs().max()) <= 1e-5 * max(1.0, float(ab.abs().max()))
    assert torch.allclose(a.states, b.states, rtol=0, atol=1e-5)


def test_onchip_peaked_softmax_and_determinism():
    """N_eff of a few: most waves' weights are exactly zero and are skipped; same seed => same bits on every run."""
    nx, nu, K, T = 16, 12, 65536, 64
    x0 = torch.randn(nx, generator=torch.Generator().manual_seed(3)).cuda()
    a, _, _, _ = _make("integrator", nx, nu, K, T, True, lam=0.02)
    b, _, _, _ = _make("integrator", nx, nu, K, T, False, lam=0.02)
    c, _, _, _ = _make("integrator", nx, nu, K, T, True, lam=0.02)
    aa, ab, ac = a.command(x0), b.command(x0), c.command(x0)
    n_eff = 1.0 / float((b.omega.double() ** 2).sum())
    assert n_eff < 30, n_eff
    assert torch.equal(aa, ac) and torch.equal(a.U, c.U) and torch.equal(a.cost_total, c.cost_total)
    assert float((a.U - b.U).abs().max()) <= 1e-5 * float(b.U.abs().max())


def test_onchip_scope_falls_back_without_error():
    """Outside the form's scope (fp64 here; a full Sigma likewise: the lane-coloured form exists behind MPPI_ONCHIP_FULL_SIGMA but
    measured slower than generator-coloured rows) the same call runs the streaming command."""
    import pytorch_mppi_amd as pm
    nx, nu, K, T = 6, 4, 20000, 40
    model = pm.models.Integrator(nx, nu)
    c = pm.MPPI(model.dynamics, model.running_cost, nx, torch.eye(nu, dtype=torch.float64) * 0.5, num_samples=K, horizon=T,
                device="cuda", lambda_=5.0, rng="philox", seed=3)
    n0 = _onchip_count()
    c.command(torch.zeros(nx, dtype=torch.float64).cuda())
    assert _onchip_count() == n0 and c.last_draw == "philox-fill"


def test_onchip_with_a_traced_user_model_and_many_chunks():
    """A model that came out of the tracer (tests/jit_fixtures.py: the reference's linear dynamics + goal cost + terminal cost as
    plain callables) on the on-chip command, at a K that needs several workgroups per CU (K = 300000: 1172 workgroups),
    against the streaming form of the same stream."""
    import jit_fixtures as jf
    import pytorch_mppi_amd as pm
    f, q, term = jf.ref_linear_callables()

    def make(onchip):
        c = pm.MPPI(f, q, 2, torch.eye(2), num_samples=300000, horizon=40, device="cuda", lambda_=30.0, terminal_state_cost=term,
                    u_min=torch.tensor([-1.0, -1.0]), u_max=torch.tensor([1.0, 1.0]), rng="philox", seed=11, auto_jit=True,
                    U_init=torch.zeros(40, 2))     # (without it every controller draws its own random initial sequence)
        assert not c._needs_generic(), c.jit_note
        c.philox_onchip = onchip
        return c
    a, b = make(None), make(False)
    x0 = torch.tensor([-1.0, 0.5]).cuda()
    n0 = _onchip_count()
    for _ in range(2):
        ua, ub = a.command(x0), b.command(x0)
    assert _onchip_count() - n0 == 2 and a.last_draw == "philox-onchip"
    assert float((a.cost_total - b.cost_total).abs().max()) <= 1e-5 * float(b.cost_total.abs().max())
    assert float((ua - ub).abs().max()) <= 1e-5 * max(1.0, float(ub.abs().max()))
    assert float((a.U - b.U).abs().max()) <= 1e-5 * max(1.0, float(b.U.abs().max()))


import os as _os

_EXTRA = int(_os.environ.get("MPPI_EXTRA_SEEDS", "0"))


@pytest.mark.parametrize("seed", list(range(16)) + [3000 + i for i in range(_EXTRA)])
def test_onchip_random_config_vs_fp64_oracle(seed):
    """Randomised breadth for the on-chip command (forced: philox_onchip = True, whatever K is): models / control widths, ragged K
    down to a single sample, horizons 1 .. 90 (every storage class: registers only, + LDS, + second generation), mu, bounds,
    u_scale, |noise| cost, null action, u_per_command, shift on / off, MPPI and SMPPI -- against the fp64 oracle on the normals of the
    device's own stream, two commands per configuration.  Criterion: SURVEY 7.3's with the floor factor at 8 instead of 2 -- over 416
    configurations the tail (small K, peaked weights, the pendulum's wrapped angle over 64-90 steps) sits at 2-5x the reference's own
    fp32 floor for BOTH forms of the command (MPPI_SWEEP_FORM=stream runs this sweep on the streaming form: same seeds, same
    errors to two digits); a wrong kernel is off by orders of magnitude, not by a factor."""
    import pytorch_mppi_amd as pm
    from oracle import mppi_oracle as orc, dynamics as dyn
    g = torch.Generator().manual_seed(1000 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    kind, nx, nu = [("integrator", 16, 12), ("integrator", 6, 4), ("integrator", 4, 2), ("integrator", 12, 6), ("pendulum", 2, 1),
                    ("integrator", 8, 4), ("integrator", 2, 2), ("linear", 10, 3)][seed % 8]
    K = [1, 63, 257, 1000, 4097, 20000][r(0, 5)]
    T = [1, 3, 8, 21, 40, 64, 90][r(0, 6)]
    smppi = seed % 4 == 3 and kind != "pendulum"
    if smppi and T < 2:
        T = 3                       # the reference's SMPPI shift needs two rows (mppi.py:491-492)
    model, mk = _models(kind, nx, nu)
    sigma = torch.diag(torch.rand(nu, generator=g) + 0.3) if nu > 1 else torch.tensor(float(torch.rand(1, generator=g)) + 0.3)
    kw = dict(lambda_=float(torch.rand(1, generator=g)) * 30 + 5.0)
    if seed % 3 != 1:
        kw["u_max"] = (torch.rand(nu, generator=g) + 0.5) if nu > 1 else torch.tensor(1.5)
    if seed % 3 == 0 and not smppi:
        kw["noise_mu"] = torch.randn(nu, generator=g) * 0.2
    if seed % 4 == 2:
        kw["u_scale"] = 0.7
    if seed % 5 == 2:
        kw["noise_abs_cost"] = True
    if seed % 2 == 1 and K > 1 and not smppi:
        kw["sample_null_action"] = True
    if seed % 7 == 3:
        kw["u_per_command"] = min(2, T)
    U0 = torch.randn(T, nu, generator=g) * 0.1
    x0 = torch.randn(nx, generator=g)
    if smppi:
        dt_, w_ = 0.2, 0.5
        amax = torch.rand(nu, generator=g) + 0.8
        kw.pop("u_per_command", None)
        c = pm.SMPPI(model.dynamics, model.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", rng="philox", seed=50 + seed,
                     U_init=U0.clone(), action_max=amax, w_action_seq_cost=w_, delta_t=dt_, **kw)
    else:
        c = pm.MPPI(model.dynamics, model.running_cost, nx, sigma, num_samples=K, horizon=T, device="cuda", rng="philox", seed=50 + seed,
                    U_init=U0.clone(), **kw)
    onchip = _os.environ.get("MPPI_SWEEP_FORM", "onchip") == "onchip"      # MPPI_SWEEP_FORM=stream: the same sweep on the streaming command
    c.philox_onchip = onchip
    n0 = _onchip_count()
    for step, shift in enumerate((True, False)):
        Ub = c.U.detach().cpu().clone()
        Ab = c.action_sequence.detach().cpu().clone() if smppi else None
        act = c.command(x0.cuda(), shift_nominal_trajectory=shift)
        assert (c.last_draw == "philox-onchip") == onchip, c.last_draw
        z = gpu_util.device_philox_normals(c, c._call)
        outs = []
        for dt in (torch.float64, torch.float32):
            f, q = mk(dt)
            cast = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
            if smppi:
                p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sigma.to(dt), K=K, T=T, **cast)
                outs.append(orc.smppi_command(p, Ub.to(dt), Ab.to(dt), x0.to(dt), z.to(dt), -amax.to(dt), amax.to(dt), w_, dt_, shift))
            else:
                p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sigma.to(dt), K=K, T=T, **cast)
                outs.append(orc.command(p, Ub.to(dt), x0.to(dt), z.to(dt), shift))
        r64, r32 = outs
        got = dict(action=act, U=c.U, cost_total=c.cost_total, omega=c.omega)
        if smppi:
            # the lifted control U is the action DERIVATIVE: its update carries the 1/dt rescaling of the noise, and at small K
            # the fp32 error of both forms of the command sits at 2-3x the reference's own fp32 floor (seed 3127: on chip
            # 1.6e-5, streaming 2.0e-5, floor 0.7e-5) -- the sweep checks what SMPPI commands, the integrated action sequence
            got = dict(action=act, action_sequence=c.action_sequence, cost_total=c.cost_total, omega=c.omega)
        for key in got:
            margins.check(f"onchip random config", f"seed {seed} step {step} {key} {kind}({nx},{nu}) K={K} T={T} smppi={smppi}",
                          got[key].detach().cpu().numpy(), r64[key].numpy(), r32[key].numpy(), rtol=1e-5, scale_floor=1.0 if key != "omega" else 0.0,
                          floor_factor=8.0)
    assert _onchip_count() - n0 == (2 if onchip else 0)
