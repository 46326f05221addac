"""Plain torch callables through the UNCHANGED constructor call run fused: `MPPI(dynamics, running_cost,
nx, sigma, device="cuda")` traces them (pytorch_mppi_amd/trace.py), compiles the functor (jit.compile_model) and takes the
K1 path.  The reference's own test callables (restated in tests/jit_fixtures.py after /root/reference/tests/pendulum.py:30-60
and tests/test_mppi.py:25-51), an nn.Module, against the fp64 oracle and against the callback path on the same draw."""
import time

import numpy as np
import pytest
import torch

import jit_fixtures as jf
import margins

pytestmark = pytest.mark.gpu


def _pair(f, q, nx, sigma, dtype, K, T, term=None, **kw):
    import pytorch_mppi_amd as pm
    g = torch.Generator().manual_seed(2)
    nu = 1 if sigma.dim() == 0 else sigma.shape[0]
    U0 = (torch.randn(T, nu, generator=g) * 0.3).to(dtype)
    mk = lambda auto: pm.MPPI(f, q, nx, sigma.to(dtype), num_samples=K, horizon=T, device="cuda", lambda_=kw.get("lambda_", 1.0),
                              terminal_state_cost=term, U_init=U0.clone(), auto_jit=auto,
                              **{k: v for k, v in kw.items() if k != "lambda_"})
    return mk(True), mk(False), U0, nu


def _best_batch_ms(c, x0, n, batches):
    """ms per command, best of `batches` batches of n: a one-off host stall (the caching allocator trimming after other tests'
    large buffers was measured at ~85 ms) must not decide a timing bound"""
    best = float("inf")
    for _ in range(batches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            c.command(x0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


def _oracle(f, q, nx, sigma, K, T, lam, U0, x0, z, term=None, **kw):
    from oracle import mppi_oracle as orc
    out = []
    for dt in (torch.float64, torch.float32):
        cast = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
        p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=sigma.to(dt), K=K, T=T, lambda_=lam, terminal_state_cost=term, **cast)
        out.append(orc.command(p, U0.to(dt), x0.to(dt), z.to(dt), True))
    return out


@pytest.mark.parametrize("dtype,K,T", [(torch.float64, 100, 15), (torch.float32, 8192, 32)])
def test_reference_pendulum_callables_run_fused(dtype, K, T):
    f, q = jf.ref_pendulum_callables()
    sigma = torch.tensor(10.0)
    kw = dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), lambda_=1.0)
    a, b, U0, nu = _pair(f, q, 2, sigma, dtype, K, T, **kw)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    assert b._needs_generic()
    # (numpy ufuncs on tensors only work on the host: on the callback path these callables raise for device tensors, in the
    # reference too -- traced, they run on the GPU.  The comparison below is with the oracle, which calls them on the host.)
    x0 = torch.tensor([math_pi(), 1.0])
    z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(8)).to(dtype)
    a.inject_noise(z)
    ua = a.command(x0.to(dtype).cuda())
    r64, r32 = _oracle(f, q, 2, sigma, K, T, 1.0, U0, x0, z, u_min=kw["u_min"], u_max=kw["u_max"])
    for name, got in (("action", ua), ("U", a.U), ("cost_total", a.cost_total)):
        if dtype == torch.float64:
            s = max(1.0, float(r64[name].abs().max()))
            assert float((got.cpu() - r64[name]).abs().max()) <= 1e-9 * s, name
        else:
            margins.check("from_torch/pendulum_f32", name, got.detach().cpu().numpy(), r64[name].numpy(), r32[name].numpy(), rtol=1e-5)
    with pytest.raises(TypeError):
        b.command(x0.to(dtype).cuda())


def math_pi():
    import math
    return math.pi


def test_reference_linear_dynamics_with_terminal_cost_run_fused():
    f, q, term = jf.ref_linear_callables()
    sigma = torch.eye(2, dtype=torch.float64)
    K, T = 500, 20
    a, b, U0, nu = _pair(f, q, 2, sigma, torch.float64, K, T, term=term, lambda_=1.0)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    x0 = torch.zeros(2, dtype=torch.float64)
    z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    a.inject_noise(z)
    b.inject_noise(z)
    ua, ub = a.command(x0.cuda()), b.command(x0.cuda())
    r64, _ = _oracle(f, q, 2, sigma, K, T, 1.0, U0, x0, z, term=term)
    for name, got in (("action", ua), ("U", a.U), ("cost_total", a.cost_total)):
        assert float((got.cpu() - r64[name]).abs().max()) <= 1e-9 * max(1.0, float(r64[name].abs().max())), name
    assert float((ua - ub).abs().max()) <= 1e-9
    assert torch.allclose(a.states, b.states, rtol=0, atol=1e-9)


def test_nn_module_dynamics_run_fused():
    f, q = jf.small_mlp_callables()
    sigma = torch.eye(2, dtype=torch.float64) * 0.5
    K, T = 700, 12
    a, b, U0, nu = _pair(f, q, 4, sigma, torch.float64, K, T, lambda_=2.0)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    x0 = torch.linspace(-1, 1, 4, dtype=torch.float64)
    z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(10), dtype=torch.float64)
    a.inject_noise(z)
    b.inject_noise(z)
    ua, ub = a.command(x0.cuda()), b.command(x0.cuda())
    assert float((ua - ub).abs().max()) <= 1e-9 and float((a.cost_total - b.cost_total).abs().max()) <= 1e-9 * float(b.cost_total.abs().max())


def test_untraceable_callables_stay_on_the_generic_path():
    import pytorch_mppi_amd as pm

    def f(s, a):
        return s + a if float(s.sum()) > 0 else s - a       # data-dependent control flow

    c = pm.MPPI(f, lambda s, a: (s ** 2).sum(-1), 2, torch.eye(2), num_samples=64, horizon=5, device="cuda", auto_jit=True)
    assert c._needs_generic() and c.jit_note.startswith("generic path"), c.jit_note
    c.command(torch.zeros(2).cuda())


def test_traced_pendulum_command_time_at_c2_size():
    """8192 x 32 through plain callables: the single-launch fused command, not the ~3 ms callback loop"""
    import pytorch_mppi_amd as pm
    f, q = jf.ref_pendulum_callables()
    c = pm.MPPI(f, q, 2, torch.tensor(10.0), num_samples=8192, horizon=32, device="cuda", lambda_=1.0,
                u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng="philox", seed=1, auto_jit=True)
    assert not c._needs_generic()
    x0 = torch.tensor([math_pi(), 1.0]).cuda()
    for _ in range(10):
        c.command(x0)
    ms = _best_batch_ms(c, x0, 50, 4)
    margins.record("from_torch/pendulum_c2_size", "ms_per_command", ms, None, 0.05, "plain torch callables, traced; bound 0.05 ms")
    assert ms <= 0.05, ms


def test_in_place_update_of_a_captured_tensor_sends_the_controller_back_to_the_callables():
    """the functor holds the VALUES of the tensors the callables read; once one is written in place the fused model would
    compute yesterday's dynamics -- the controller notices (version counter) and runs the callables again"""
    import pytorch_mppi_amd as pm
    f, q, B = jf.watched_linear_callables()
    mk = lambda auto: pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.float64), num_samples=256, horizon=6, device="cuda",
                              U_init=torch.zeros(6, 2, dtype=torch.float64), auto_jit=auto)
    a, b = mk(True), mk(False)
    assert not a._needs_generic()
    x0 = torch.ones(2, dtype=torch.float64).cuda()
    z = torch.randn(256, 6, 2, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    for c in (a, b):
        c.inject_noise(z)
    assert float((a.command(x0) - b.command(x0)).abs().max()) <= 1e-9
    B[0, 1] = 0.5                                   # the dynamics change under the controller
    for c in (a, b):
        c.inject_noise(z)
    ua, ub = a.command(x0), b.command(x0)
    assert a._needs_generic() and a.jit_note.startswith("generic path")
    assert float((ua - ub).abs().max()) <= 1e-9
    # ... while the functor is traced again beside the loop, the matrix as run-time parameters this time
    assert a.wait_for_jit(300.0) and not a._needs_generic() and a._model._n_params == 4, a.jit_note
    B[1, 1] = -0.25
    for c in (a, b):
        c.U = torch.zeros(6, 2, dtype=torch.float64, device="cuda")
        c.inject_noise(z)
    assert float((a.command(x0) - b.command(x0)).abs().max()) <= 1e-9 and not a._needs_generic()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_learned_dynamics_stay_fused_while_the_network_is_trained(dtype):
    """/root/reference/tests/pendulum_approximate.py: the dynamics are a trainable 3 -> 32 -> 32 -> 2 tanh network that the
    user's loop retrains between commands.  Its parameters are run-time inputs of the traced functor (one rolled copy of the
    ~3700-operation step, weights through scalar loads): the controller stays on the fused kernels across optimizer steps
    and commands what the callback loop commands on the same draw -- nothing is recompiled."""
    f, q, net = jf.approx_pendulum_callables(dtype=dtype)
    net.cuda()
    K, T = (1000, 30) if dtype == torch.float64 else (4096, 30)          # (the reference example's K x T)
    kw = dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), lambda_=1.0)
    a, b, U0, nu = _pair(f, q, 2, torch.tensor(1.0), dtype, K, T, **kw)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    assert a._model.heavy and a._model.wide and a._model._n_params == 1250 and b._needs_generic()
    model = a._model
    x0 = torch.tensor([2.5, -0.8], dtype=dtype).cuda()
    gen = torch.Generator().manual_seed(21)
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    for rnd in range(3):
        z = torch.randn(K, T, nu, generator=gen, dtype=torch.float64).to(dtype)
        for c in (a, b):
            c.inject_noise(z)
        ua, ub = a.command(x0), b.command(x0)
        assert a._model is model and not a._needs_generic(), a.jit_note
        s = max(1.0, float(b.cost_total.abs().max()))
        bad = ((a.cost_total - b.cost_total).abs() > tol * s).double().mean().item()
        if dtype == torch.float64:
            assert bad == 0.0, rnd
            assert float((ua - ub).abs().max()) <= tol * max(1.0, float(ub.abs().max())), (rnd, ua, ub)
        else:
            # fp32 against fp32: the angle wrap makes the learned dynamics discontinuous at +-pi, so a sample that passes
            # within rounding of the seam may take the other branch in the other implementation -- isolated samples only
            assert bad <= 2e-3, (rnd, bad)
            assert float((ua - ub).abs().max()) <= 5e-3 * max(1.0, float(ub.abs().max())), (rnd, ua, ub)
        if rnd == 0:
            jf.train_a_little(net, steps=3, seed=rnd)                    # optimizer.step(): in-place updates
        elif rnd == 1:
            for p_ in net.parameters():                                  # frozen for control (pendulum_approximate.py:150-170)
                p_.requires_grad_(False)
            net.load_state_dict({k: v * 0.9 for k, v in net.state_dict().items()})


def test_learned_dynamics_command_time():
    """the same learned dynamics at the C2 problem size (8192 x 32, fp32): one rolled copy of the step in K1"""
    import pytorch_mppi_amd as pm
    f, q, net = jf.approx_pendulum_callables(dtype=torch.float32)
    net.cuda()
    mk = lambda auto: pm.MPPI(f, q, 2, torch.tensor(1.0), num_samples=8192, horizon=32, device="cuda", lambda_=1.0,
                              u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), rng="philox", seed=1, auto_jit=auto)
    x0 = torch.tensor([2.5, -0.8]).cuda()
    out = {}
    for name, auto, n in (("fused", True, 100), ("callbacks", False, 10)):
        c = mk(auto)
        assert c._needs_generic() == (not auto)
        for _ in range(3):
            c.command(x0)
        out[name] = _best_batch_ms(c, x0, n // 4, 4)
    margins.record("from_torch/learned_pendulum_c2_size", "ms_per_command", out["fused"], None, 0.06,
                   "trainable 3-32-32-2 tanh network traced with run-time parameters, its layers on the matrix cores "
                   "(sixteen samples per wave); callback loop: %.3f ms" % out["callbacks"])
    # (0.20 ms with one lane per sample: round 3; 0.0545-0.0558 measured in round 4 -- the bound leaves the 10 % by which the
    #  boxes of the pool differ from one another: a regression guard, the number itself is in profiles/r04_final2_learned_bench.txt)
    assert out["fused"] <= 0.066 and out["fused"] * 20 <= out["callbacks"], out


def test_wider_operator_vocabulary_runs_fused():
    """GELU (erf), ELU (expm1), ReLU, atan by item assignment, Huber cost, norm, a logical mask: traced, compiled, and equal to
    the callback loop on the same draw (fp64)."""
    f, q, net = jf.zoo_callables()
    sigma = torch.eye(2, dtype=torch.float64) * 0.5
    K, T = 900, 14
    a, b, U0, nu = _pair(f, q, 4, sigma, torch.float64, K, T, lambda_=1.5)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    x0 = torch.linspace(-1, 1, 4, dtype=torch.float64)
    z = torch.randn(K, T, nu, generator=torch.Generator().manual_seed(12), dtype=torch.float64)
    a.inject_noise(z)
    b.inject_noise(z)
    ua, ub = a.command(x0.cuda()), b.command(x0.cuda())
    assert float((ua - ub).abs().max()) <= 1e-9 and float((a.cost_total - b.cost_total).abs().max()) <= 1e-9 * float(b.cost_total.abs().max())


@pytest.mark.parametrize("variant", ["kmppi", "smppi", "states_null_action_terminal"])
def test_heavy_traced_model_under_the_other_controllers(variant):
    """the one-copy-of-the-step stream (`rollout_stream_heavy`) behind everything that reaches K1: KMPPI (two-launch form: the
    interpolated actions arrive as rows), SMPPI (base sequence, smoothness cost), the `states` output, the null-action row and
    a terminal cost -- against the callback loop on the same draw, fp64."""
    import pytorch_mppi_amd as pm
    f, q, net = jf.approx_pendulum_callables()
    net.cuda()
    K, T = 600, 20
    sigma = torch.tensor(0.8, dtype=torch.float64)
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=1.0, U_init=torch.zeros(T, 1, dtype=torch.float64))
    shape = (K, T, 1)
    if variant == "kmppi":
        mk = lambda auto: pm.KMPPI(f, q, 2, sigma, num_support_pts=6, u_min=torch.tensor(-2.0, dtype=torch.float64),
                                   u_max=torch.tensor(2.0, dtype=torch.float64), auto_jit=auto, **kw)
        shape = (K, 6, 1)
        keys = ("U", "theta", "cost_total")
    elif variant == "smppi":
        mk = lambda auto: pm.SMPPI(f, q, 2, sigma, action_max=torch.tensor([1.5], dtype=torch.float64), w_action_seq_cost=0.7,
                                   delta_t=0.1, auto_jit=auto, **kw)
        keys = ("U", "action_sequence", "cost_total")
    else:
        mk = lambda auto: pm.MPPI(f, q, 2, sigma, terminal_state_cost=jf.approx_terminal_cost, sample_null_action=True, auto_jit=auto, **kw)
        keys = ("U", "cost_total", "states")       # (the visited states are kept when a terminal cost is set, mppi.py:307-310)
    a, b = mk(True), mk(False)
    assert a.jit_note.startswith("fused") and a._model.heavy and not a._needs_generic(), a.jit_note
    assert b._needs_generic()
    x0 = torch.tensor([2.0, -0.5], dtype=torch.float64).cuda()
    gen = torch.Generator().manual_seed(31)
    for step in range(2):
        z = torch.randn(*shape, generator=gen, dtype=torch.float64)
        for c in (a, b):
            c.inject_noise(z)
        ua, ub = a.command(x0), b.command(x0)
        assert float((ua - ub).abs().max()) <= 1e-9 * max(1.0, float(ub.abs().max())), (variant, step)
        for k in keys:
            ga, gb = getattr(a, k), getattr(b, k)
            assert ga.shape == gb.shape, (variant, k, ga.shape, gb.shape)
            assert float((ga - gb).abs().max()) <= 1e-9 * max(1.0, float(gb.abs().max())), (variant, step, k)


def test_reference_schedule_indexed_by_the_timestep_runs_fused():
    """trajectory tracking under step_dependent_dynamics=True: `ref[t]` / `gain[t]` lookups of constant tensors are constant
    tables in the functor (read with the wave-uniform timestep); against the callback loop on the same draw, fp64"""
    import pytorch_mppi_amd as pm
    f, q = jf.tracking_callables()
    K, T = 700, 24
    kw = dict(num_samples=K, horizon=T, device="cuda", lambda_=1.0, step_dependent_dynamics=True,
              U_init=torch.zeros(T, 2, dtype=torch.float64))
    a = pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.float64) * 0.6, auto_jit=True, **kw)
    b = pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.float64) * 0.6, auto_jit=False, **kw)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    x0 = torch.tensor([0.3, 0.8], dtype=torch.float64).cuda()
    gen = torch.Generator().manual_seed(41)
    for step in range(2):
        z = torch.randn(K, T, 2, generator=gen, dtype=torch.float64)
        for c in (a, b):
            c.inject_noise(z)
        ua, ub = a.command(x0), b.command(x0)
        assert float((ua - ub).abs().max()) <= 1e-9 and float((a.cost_total - b.cost_total).abs().max()) <= 1e-9 * float(b.cost_total.abs().max())


# ---------------------------------------------------------------------------------------------------------------------
# traced callables against the live ones: the reference calls the callables on every command
# (mppi.py:314, :318; tests/smooth_mppi.py:54-58 reads `self.goal` live), so state changed between commands must
# reach a controller that runs a traced functor -- on the very next command
# ---------------------------------------------------------------------------------------------------------------------
def _retarget_oracle(f, q, K, T, U, x0, z):
    from oracle import mppi_oracle as orc
    p = orc.Problem(dynamics=f, running_cost=q, nx=2, noise_sigma=torch.eye(2, dtype=torch.float64), K=K, T=T, lambda_=1.0)
    return orc.command(p, U, x0, z, True)


def test_rebound_goal_changed_gain_and_rewritten_matrix_reach_the_fused_controller():
    import pytorch_mppi_amd as pm
    f, q, holder = jf.retargeted_callables()
    K, T = 512, 8
    U0 = torch.zeros(T, 2, dtype=torch.float64)
    a = pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.float64), num_samples=K, horizon=T, device="cuda", U_init=U0.clone(), auto_jit=True)
    assert a.jit_note.startswith("fused") and not a._needs_generic(), a.jit_note
    x0 = torch.tensor([0.3, -0.2], dtype=torch.float64)
    gen = torch.Generator().manual_seed(21)

    def step(tag):
        """one command on a fresh draw, checked against the oracle that calls the LIVE callables (fp64, 1e-9)"""
        z = torch.randn(K, T, 2, generator=gen, dtype=torch.float64)
        U_before = a.U.detach().cpu().clone()
        a.inject_noise(z)
        ua = a.command(x0.cuda())
        r = _retarget_oracle(f, q, K, T, U_before, x0, z)
        for name, got in (("action", ua), ("U", a.U), ("cost_total", a.cost_total)):
            s = max(1.0, float(r[name].abs().max()))
            assert float((got.cpu() - r[name]).abs().max()) <= 1e-9 * s, (tag, name)

    step("as traced")
    step("as traced, again")
    m0 = a._model
    assert a._jit_spot_checks >= 1 and a._jit_retraces == 0 and not a._needs_generic()
    # 1. the goal tensor is REBOUND and a Python float changed: the very next command follows the new cost
    q.goal = torch.tensor(jf.RETARGET_GOAL_2, dtype=torch.float64)
    q.gain = jf.RETARGET_GAIN_2
    step("goal rebound + gain changed")
    assert a._jit_retraces == 1 and "RetargetedCost.goal" in " ".join(repr(p_) for p_ in a._jit_dynamic)
    assert a.wait_for_jit(300.0) and a._model is not m0 and not a._needs_generic(), a.jit_note
    m1 = a._model
    assert m1._n_params == 2 and "p[0]" in m1._code["cost"]        # the goal is a run-time parameter now
    step("new functor")
    # 2. the next goals cost one small copy each: same kernels
    for goal in ([0.25, 0.5], [2.0, -2.0]):
        q.goal = torch.tensor(goal, dtype=torch.float64)
        step(f"goal {goal}")
        assert a._model is m1 and not a._needs_generic()
    q.goal[1] = 0.125                                                # ... written in place too
    step("goal written in place")
    assert a._model is m1 and a._jit_retraces == 1
    # 3. a matrix the dynamics read is written IN PLACE (version counter): promoted as well
    holder["B"][0, 1] = 0.15
    step("B written in place")
    assert a._jit_retraces == 2 and a.wait_for_jit(300.0) and a._model._n_params == 6
    step("B as parameters")
    holder["B"] = holder["B"] * 0.5                                  # rebound this time
    m2 = a._model
    step("B rebound")
    assert a._model is m2 and not a._needs_generic()


def test_writes_the_watch_cannot_see_are_caught_by_the_spot_check():
    """`tensor.data[...] = v` moves no version counter and no pointer: the periodic functor-against-callables check on the
    device (MPPI._spot_check) is what notices"""
    import pytorch_mppi_amd as pm
    f, q, B = jf.watched_linear_callables()
    mk = lambda auto: pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.float64), num_samples=256, horizon=6, device="cuda",
                              U_init=torch.zeros(6, 2, dtype=torch.float64), auto_jit=auto)
    a, b = mk(True), mk(False)
    assert not a._needs_generic()
    a._jit_check_every, a._jit_check_share = 4, 0.0      # (exactly every 4th command: no stretching for cheap commands)
    x0 = torch.ones(2, dtype=torch.float64).cuda()
    gen = torch.Generator().manual_seed(5)

    def both():
        z = torch.randn(256, 6, 2, generator=gen, dtype=torch.float64)
        a.U = b.U.clone()
        for c in (a, b):
            c.inject_noise(z)
        return float((a.command(x0) - b.command(x0)).abs().max())
    for _ in range(6):
        assert both() <= 1e-9
    n0 = a._jit_spot_checks
    assert n0 >= 2 and not a._needs_generic()
    B.data[1, 0] = 0.75                               # invisible to version counters
    worst = [both() for _ in range(4)]                # at most three commands until the next check ...
    assert a._jit_spot_checks > n0 and a._jit_retraces == 1
    assert both() <= 1e-9                             # ... then the callables again, and the new functor later
    assert max(worst) > 1e-6                          # (the window the spot-check interval leaves: documented in DESIGN.md 2b)
    assert a.wait_for_jit(300.0)
    assert both() <= 1e-9 and not a._needs_generic()


def test_replaced_module_keeps_the_kernels():
    import pytorch_mppi_amd as pm
    f, q, holder, mk_net = jf.swappable_net_callables()
    mk = lambda auto: pm.MPPI(f, q, 2, torch.tensor(0.5, dtype=torch.float64), num_samples=300, horizon=7, device="cuda",
                              U_init=torch.zeros(7, 1, dtype=torch.float64), auto_jit=auto)
    holder["net"].cuda()
    a, b = mk(True), mk(False)
    assert not a._needs_generic(), a.jit_note
    m = a._model
    x0 = torch.tensor([0.4, -0.1], dtype=torch.float64).cuda()
    gen = torch.Generator().manual_seed(6)

    def both():
        z = torch.randn(300, 7, 1, generator=gen, dtype=torch.float64)
        a.U = b.U.clone()
        for c in (a, b):
            c.inject_noise(z)
        return float((a.command(x0) - b.command(x0)).abs().max())
    assert both() <= 1e-9
    holder["net"] = mk_net(99).cuda()                 # same architecture, other weights
    assert both() <= 1e-9 and a._model is m and not a._needs_generic() and a._jit_retraces == 0
    with torch.no_grad():
        holder["net"][0].weight.data.mul_(0.5)        # a write through .data: re-gathered by the spot-check
    a._jit_check_every, a._jit_check_share, a._jit_next_check = 1, 0.0, 0
    assert both() <= 1e-9 and a._model is m


# ---------------------------------------------------------------------------------------------------------------------
# dense layers of traced models on the matrix cores (VERDICT r03 missing #2; csrc/mlp_wide.hpp): sixteen samples per wave,
# v_mfma_f32_16x16x4_f32 tiles for the layers, the scalar part of the functor replicated in the four lane groups
# ---------------------------------------------------------------------------------------------------------------------
def _wide_pair(f, q, nx, sigma, K, T, **kw):
    """(matrix-core kernel, one-lane-per-sample kernels of the SAME functor, callback loop)"""
    import pytorch_mppi_amd as pm
    nu = 1 if sigma.dim() == 0 else sigma.shape[0]
    mk = lambda auto: pm.MPPI(f, q, nx, sigma.float(), num_samples=K, horizon=T, device="cuda", U_init=torch.zeros(T, nu), auto_jit=auto, **kw)
    a, s, b = mk(True), mk(True), mk(False)
    assert a._model is not None and a._model.wide and not a._needs_generic(), a.jit_note
    import copy
    s._model = copy.copy(a._model)           # the same compiled functor, told to stay on the one-lane-per-sample kernels
    s._model.use_wide = False
    s._model.invalidate()
    s._problem_cache.clear()
    return a, s, b, nu


@pytest.mark.parametrize("K,T", [(8192, 32), (1000, 30), (77, 9)])
def test_traced_network_on_the_matrix_cores_matches_the_per_lane_kernels_and_the_callbacks(K, T):
    """the reference's learned pendulum (tests/pendulum_approximate.py:47-67: 3 -> 32 -> 32 -> 2 tanh, action clamped, angle
    wrapped): cost_total of the wide kernel against the one-lane-per-sample kernels of the same functor (same fp32 arithmetic up to
    the order of the sums) and against the callback loop; ragged K (not a multiple of 16 / 64), the null action, a terminal cost"""
    f, q, net = jf.approx_pendulum_callables(dtype=torch.float32)
    net.cuda()
    kw = dict(u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), lambda_=1.0, sample_null_action=True, terminal_state_cost=jf.approx_terminal_cost)
    a, s, b, nu = _wide_pair(f, q, 2, torch.tensor(1.0), K, T, **kw)
    x0 = torch.tensor([2.5, -0.8]).cuda()
    gen = torch.Generator().manual_seed(K)
    for rnd in range(2):
        z = torch.randn(K, T, nu, generator=gen)
        for c in (a, s, b):
            c.U = torch.zeros(T, nu, device="cuda") if rnd == 0 else a.U.clone()
            c.inject_noise(z)
        ua, us, ub = a.command(x0), s.command(x0), b.command(x0)
        sc = max(1.0, float(b.cost_total.abs().max()))
        for other, name in ((s, "per-lane kernels"), (b, "callbacks")):
            bad = ((a.cost_total - other.cost_total).abs() > 2e-4 * sc).double().mean().item()
            # (the angle wrap makes the dynamics discontinuous at +-pi: isolated samples may take the other branch)
            assert bad <= 2e-3, (name, rnd, bad)
        assert float((ua - us).abs().max()) <= 5e-3 and float((ua - ub).abs().max()) <= 5e-3, (ua, us, ub)
        # visited states of the wide kernel (stored by the lane group that owns the sample)
        st_a, st_b = a.states, b.states
        close = ((st_a - st_b).abs().amax(dim=(0, 2, 3)) <= 1e-3 * max(1.0, float(st_b.abs().max()))).double().mean().item()
        assert close >= 0.995, close
        if rnd == 0:
            jf.train_a_little(net, steps=2, seed=3)          # the weights are run-time parameters of the wide kernel too


def test_traced_network_with_mixed_activations_and_a_wide_state():
    """a network whose activations are not all one-operand functions (GELU reads its argument twice: that layer stands alone;
    ReLU and the sigmoid fuse), nx = 4, nu = 2, 6 outputs; fp64 controllers keep the per-lane kernels, fp32 ones take the matrix cores"""
    import pytorch_mppi_amd as pm
    f, q, net = jf.relu_net_callables()
    net.cuda()
    sigma = torch.eye(2) * 0.4
    a, s, b, nu = _wide_pair(f, q, 4, sigma, 900, 14, lambda_=2.0)
    code = a._model._code
    assert "mlp_single(ml0_0" in code["step"] and "mlp_first(ml1_0" in code["step"] and "mlp_mid(ml2_1" in code["step"] and "mlp_last(ml3_1" in code["step"]
    x0 = torch.linspace(-0.5, 0.5, 4).cuda()
    z = torch.randn(900, 14, 2, generator=torch.Generator().manual_seed(12))
    for c in (a, s, b):
        c.inject_noise(z)
    ua, us, ub = a.command(x0), s.command(x0), b.command(x0)
    sc = max(1.0, float(b.cost_total.abs().max()))
    assert float((a.cost_total - s.cost_total).abs().max()) <= 1e-4 * sc and float((a.cost_total - b.cost_total).abs().max()) <= 1e-4 * sc
    assert float((ua - ub).abs().max()) <= 1e-4 and float((us - ub).abs().max()) <= 1e-4
    f64, q64, net64 = jf.relu_net_callables(dtype=torch.float64)
    net64.cuda()
    d = pm.MPPI(f64, q64, 4, sigma.double(), num_samples=300, horizon=10, device="cuda", auto_jit=True)
    e = pm.MPPI(f64, q64, 4, sigma.double(), num_samples=300, horizon=10, device="cuda", auto_jit=False)
    assert d._model is not None and d._model.wide                      # (fp64: the same kind of functor, fma chains per lane)
    z = torch.randn(300, 10, 2, generator=torch.Generator().manual_seed(13), dtype=torch.float64)
    for c in (d, e):
        c.U = torch.zeros(10, 2, dtype=torch.float64, device="cuda")
        c.inject_noise(z)
    assert float((d.command(x0.double()) - e.command(x0.double())).abs().max()) <= 1e-9


def test_c4_shaped_torch_callables_take_the_matrix_core_mlp_kernel():
    """VERDICT r04 missing #2 / item 4: BASELINE configs[3] written as the reference's plugin API -- a plain
    nn.Sequential(Linear(20, 256), Tanh(), Linear(256, 16)) residual model and cost sum x^2 as torch callables -- is recognised by
    the tracer (trace.match_mlp_residual) and rolled out by the hand-written split-operand matrix-core kernel
    (csrc/rollout_mlp_split.hip) instead of the exact-fp32 wide form of the traced layers: no hipcc run, the trainable weights
    are run-time parameters (retraining between commands keeps the kernel), parity against the fp64 oracle on the consumed draw
    by the C4 criterion, and the C4-sized command in <= 0.5 ms."""
    import gpu_util
    import pytorch_mppi_amd as pm
    from pytorch_mppi_amd import _native as N
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(20, 256), torch.nn.Tanh(), torch.nn.Linear(256, 16)).cuda()
    f = lambda x, u: x + 0.1 * net(torch.cat((x, u), dim=-1))
    q = lambda x, u: (x ** 2).sum(dim=-1)
    nx, nu, K, T = 16, 4, 16384, 32
    g = torch.Generator().manual_seed(2)
    U0 = torch.randn(T, nu, generator=g) * 0.05
    x0 = torch.randn(nx, generator=g)
    mk = lambda K_, T_, lam: pm.MPPI(f, q, nx, torch.eye(nu), num_samples=K_, horizon=T_, device="cuda", lambda_=lam, rng="philox", seed=11,
                                     U_init=(U0 if T_ == T else torch.zeros(T_, nu)), auto_jit="sync")
    c = mk(K, T, 1.0)
    assert c._model is not None and c._model.model_id == N.MODEL_MLP and c._model.hidden == 256 and not c._needs_generic(), c.jit_note
    c.command(x0.cuda())
    lam = float(c.cost_total.std())
    for rnd in range(2):
        c = mk(K, T, lam)
        act = c.command(x0.cuda())
        z = gpu_util.consumed_normals(c)
        net_cpu = [p_.detach().cpu() for p_ in net.parameters()]

        def f_ref(dt):
            W1, b1, W2, b2 = (w.to(dt) for w in net_cpu)
            return lambda x, u: x + 0.1 * (torch.tanh(torch.cat((x, u), dim=-1) @ W1.T + b1) @ W2.T + b2)
        outs = []
        from oracle import mppi_oracle as orc
        for dt in (torch.float64, torch.float32):
            p = orc.Problem(dynamics=f_ref(dt), running_cost=q, nx=nx, noise_sigma=torch.eye(nu, dtype=dt), K=K, T=T, lambda_=lam)
            outs.append(orc.command(p, U0.to(dt), x0.to(dt), z.to(dt), True))
        r64, r32 = outs
        for name, got in (("action", act), ("U", c.U), ("cost_total", c.cost_total), ("omega", c.omega)):
            margins.check(f"from_torch/c4-shaped callables on the split MFMA kernel round {rnd}", name, got.detach().cpu().numpy(), r64[name].numpy(),
                          r32[name].numpy(), rtol=1e-5)
        # retrain between commands: in-place parameter updates reach the kernel's blob (refresh_params), nothing is compiled
        opt = torch.optim.SGD(net.parameters(), lr=1e-2)
        loss = (net(torch.randn(64, 20, device="cuda")) ** 2).mean()
        loss.backward()
        opt.step()
    # the C4-sized command
    big = mk(65536, 64, lam)
    assert big._model.model_id == N.MODEL_MLP
    for _ in range(3):
        big.command(x0.cuda())
    xd = x0.cuda()
    ms = _best_batch_ms(big, xd, 20, 3)
    # the built-in model object of the same shape through the same loop (what bench.py's C4 line runs), for the record
    ref = pm.models.MLPResidual.random(nx, nu, 256, seed=2)
    cb = pm.MPPI(ref.dynamics, ref.running_cost, nx, torch.eye(nu), num_samples=65536, horizon=64, device="cuda", lambda_=lam, rng="philox", seed=11)
    for _ in range(3):
        cb.command(xd)
    ms_builtin = _best_batch_ms(cb, xd, 20, 3)
    margins.record("from_torch/c4-shaped callables on the split MFMA kernel", "ms_per_command", ms, None, 0.55,
                   "nn.Sequential(Linear(20,256), Tanh(), Linear(256,16)) residual + sum x^2 as torch callables, K 65536 x T 64; "
                   "the built-in models.MLPResidual through the same loop: %.4f ms" % ms_builtin)
    # (measured 0.471-0.479 inside this suite on three boxes -- the live-callable spot-check rides along with every command -- and
    #  0.462 in bench.py's loop, profiles/r05_final_*: the number the <= 0.50 claim stands on is THERE; the guard here leaves the
    #  10 % by which the boxes of the pool differ from one another, so that a slow box does not fail the suite)
    assert ms <= 0.55 and ms <= 1.12 * ms_builtin, (ms, ms_builtin)


@pytest.mark.parametrize("kernel", ["split", "exact", "valu"])
def test_mlp_model_with_a_diagonal_quadratic_cost(kernel, monkeypatch):
    """round 5: the MLP model family's running cost is sum_i q_i x_i^2 + sum_n r_n u_n^2 (weights in the blob behind res_scale; ones and
    zeros = the plain sum x^2 of BASELINE configs[3..4], same bits as before) -- in the split-operand matrix-core kernel, the exact
    fp32 one and the per-lane functor; as a built-in model object and as torch callables the tracer recognises"""
    import gpu_util
    import pytorch_mppi_amd as pm
    from pytorch_mppi_amd import _native as N
    from oracle import dynamics as dyn
    from oracle import mppi_oracle as orc
    if kernel == "exact":
        monkeypatch.setenv("MPPI_MLP_EXACT", "1")
    elif kernel == "valu":
        monkeypatch.setenv("MPPI_MLP_VALU", "1")
    nx, nu, H, K, T = 16, 4, 64, 8192, 24
    W = dyn.make_mlp_weights(nx, nu, H, seed=7)
    qs = torch.linspace(0.25, 2.0, nx)
    qc = torch.tensor([0.05, 0.1, 0.2, 0.4])
    g = torch.Generator().manual_seed(2)
    U0 = torch.randn(T, nu, generator=g) * 0.05
    x0 = torch.randn(nx, generator=g)
    built_in = pm.models.MLPResidual(*W, nx, nu, q_state=qs, q_control=qc)
    net = torch.nn.Sequential(torch.nn.Linear(nx + nu, H), torch.nn.Tanh(), torch.nn.Linear(H, nx))
    with torch.no_grad():
        for p_, w in zip(net.parameters(), W):
            p_.copy_(w)
    net.cuda()
    qsd, qcd = qs.cuda(), qc.cuda()
    f_t = lambda x, u: x + 0.1 * net(torch.cat((x, u), dim=-1))
    q_t = lambda x, u: (qsd * x ** 2).sum(dim=-1) + (qcd * u ** 2).sum(dim=-1)
    ctrls = {"built-in": lambda lam: pm.MPPI(built_in.dynamics, built_in.running_cost, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda",
                                            lambda_=lam, rng="philox", seed=5, U_init=U0.clone(), u_scale=1.5),
             "callables": lambda lam: pm.MPPI(f_t, q_t, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=lam, rng="philox",
                                             seed=5, U_init=U0.clone(), u_scale=1.5, auto_jit="sync")}
    probe = ctrls["built-in"](1.0)
    probe.command(x0.cuda())
    lam = float(probe.cost_total.std())
    outs = {}
    for name, mk in ctrls.items():
        c = mk(lam)
        assert c._model is not None and c._model.model_id == N.MODEL_MLP and not c._needs_generic(), (name, c.jit_note)
        act = c.command(x0.cuda())
        z = gpu_util.consumed_normals(c)
        res = []
        for dt in (torch.float64, torch.float32):
            f, q = dyn.make_mlp(*[w.to(dt) for w in W], q_state=qs.to(dt), q_control=qc.to(dt))
            p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=torch.eye(nu, dtype=dt), K=K, T=T, lambda_=lam, u_scale=1.5)
            res.append(orc.command(p, U0.to(dt), x0.to(dt), z.to(dt), True))
        for k_, got in (("action", act), ("U", c.U), ("cost_total", c.cost_total), ("omega", c.omega)):
            margins.check(f"mlp diagonal quadratic cost / {kernel} / {name}", k_, got.detach().cpu().numpy(), res[0][k_].numpy(), res[1][k_].numpy(), rtol=1e-5)
        outs[name] = (act, c.cost_total)
    assert torch.equal(outs["built-in"][1], outs["callables"][1]), "the same kernel on the same blob"


@pytest.mark.parametrize("H", [100, 200])
def test_mlp_of_an_odd_hidden_width_runs_on_the_matrix_core_kernel_with_padding_units(H):
    """a hidden width between the widths the matrix-core kernels are built for (64 / 128 / 256) is zero-padded to the next one
    (models.mlp_kernel_width / pad_hidden: padding units have zero weights in and out, tanh(0) = 0 adds exactly nothing) instead of
    falling to the per-lane kernel -- as a built-in model object and as torch callables; parity with the fp64 oracle of the UNPADDED
    network, and the same bits as a model somebody padded by hand"""
    import gpu_util
    import pytorch_mppi_amd as pm
    from pytorch_mppi_amd import _native as N
    from oracle import dynamics as dyn
    from oracle import mppi_oracle as orc
    nx, nu, K, T = 16, 4, 8192, 24
    Hp = 128 if H <= 128 else 256
    W = dyn.make_mlp_weights(nx, nu, H, seed=9)
    g = torch.Generator().manual_seed(4)
    U0 = torch.randn(T, nu, generator=g) * 0.05
    # (a start near the goal: with |x0| ~ 4 the cost every sample shares is ~40 lambda, and omega = exp(-(c - min) / lambda) / eta then
    #  carries 40 x the costs' fp32 rounding -- in the reference's own fp32 run as much as here, a test of nothing)
    x0 = torch.randn(nx, generator=g) * 0.3
    built_in = pm.models.MLPResidual(*W, nx, nu)
    assert built_in.hidden == Hp and built_in.hidden_units == H
    W1p, b1p, W2p = pm.models.pad_hidden(W[0], W[1], W[2], Hp)
    by_hand = pm.models.MLPResidual(W1p, b1p, W2p, W[3], nx, nu)
    net = torch.nn.Sequential(torch.nn.Linear(nx + nu, H), torch.nn.Tanh(), torch.nn.Linear(H, nx))
    with torch.no_grad():
        for p_, w in zip(net.parameters(), W):
            p_.copy_(w)
    net.cuda()
    f_t = lambda x, u: x + 0.1 * net(torch.cat((x, u), dim=-1))
    q_t = lambda x, u: (x ** 2).sum(dim=-1)
    mk = lambda f, q, lam, **kw: pm.MPPI(f, q, nx, torch.eye(nu), num_samples=K, horizon=T, device="cuda", lambda_=lam, rng="philox", seed=5,
                                         U_init=U0.clone(), **kw)
    probe = mk(built_in.dynamics, built_in.running_cost, 1.0)
    probe.command(x0.cuda())
    lam = float(probe.cost_total.std())
    outs = {}
    for name, c in (("built-in", mk(built_in.dynamics, built_in.running_cost, lam)), ("padded by hand", mk(by_hand.dynamics, by_hand.running_cost, lam)),
                    ("callables", mk(f_t, q_t, lam, auto_jit="sync"))):
        assert c._model is not None and c._model.model_id == N.MODEL_MLP and c._model.hidden == Hp and not c._needs_generic(), (name, c.jit_note)
        act = c.command(x0.cuda())
        z = gpu_util.consumed_normals(c)
        res = []
        for dt in (torch.float64, torch.float32):
            f, q = dyn.make_mlp(*[w.to(dt) for w in W])
            p = orc.Problem(dynamics=f, running_cost=q, nx=nx, noise_sigma=torch.eye(nu, dtype=dt), K=K, T=T, lambda_=lam)
            res.append(orc.command(p, U0.to(dt), x0.to(dt), z.to(dt), True))
        for k_, got in (("action", act), ("U", c.U), ("cost_total", c.cost_total), ("omega", c.omega)):
            margins.check(f"mlp hidden {H} padded to {Hp} / {name}", k_, got.detach().cpu().numpy(), res[0][k_].numpy(), res[1][k_].numpy(), rtol=1e-5)
        outs[name] = c.cost_total.clone()
    assert torch.equal(outs["built-in"], outs["padded by hand"]) and torch.equal(outs["built-in"], outs["callables"])
    # retraining the callables' network still reaches the (padded) blob
    c = mk(f_t, q_t, lam, auto_jit="sync")
    c.command(x0.cuda())
    before = c.cost_total.clone()
    with torch.no_grad():
        net[2].weight.mul_(1.5)
    c.reset()
    c.U = U0.clone().cuda()
    c.command(x0.cuda())
    assert not torch.equal(before, c.cost_total)
