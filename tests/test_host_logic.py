"""CPU: host-side mirror of the reference API (constructor resolution, attribute surface, KMPPI
operators, shard arithmetic, Philox restatement) -- no kernel is launched here."""
import numpy as np
import pytest
import torch

import pytorch_mppi_amd as pm
from pytorch_mppi_amd import models
from pytorch_mppi_amd.dist import ShardPlan, combine_records_host
from oracle import mppi_oracle as orc
from oracle import philox as oph


def _lin():
    return models.LinearGoal(torch.tensor([[1.0, 0.0], [0.0, -1.0]], dtype=torch.double),
                             torch.tensor([2.0, 2.0], dtype=torch.double))


def test_constructor_resolution_matches_reference_rules():
    m = _lin()
    sigma = torch.tensor([[1.0, 0.4], [0.4, 0.5]], dtype=torch.double)
    c = pm.MPPI(m.dynamics, m.running_cost, 2, sigma, num_samples=128, horizon=12, u_max=torch.tensor([1.5, 1.0], dtype=torch.double),
                lambda_=2.5, noise_mu=torch.tensor([0.1, -0.2], dtype=torch.double), U_init=torch.zeros(12, 2, dtype=torch.double))
    p = orc.Problem(dynamics=m.dynamics, running_cost=m.running_cost, nx=2, noise_sigma=sigma, K=128, T=12,
                    u_max=torch.tensor([1.5, 1.0], dtype=torch.double))
    assert c.nu == 2 and c.dtype == torch.double and (c.K, c.T) == (128, 12)
    assert torch.equal(c.u_min, -c.u_max) and torch.equal(c.u_min, p.u_min)       # mppi.py:112-119
    assert not c._diagonal_sigma
    assert torch.equal(c._noise_L, p.fac["chol"]) and torch.equal(c.noise_sigma_inv, p.fac["sigma_inv"])
    assert c._model is m
    d = pm.MPPI(m.dynamics, m.running_cost, 2, torch.diag(torch.tensor([1.0, 4.0], dtype=torch.double)), U_init=torch.zeros(15, 2, dtype=torch.double))
    assert d._diagonal_sigma and torch.equal(d._noise_L, torch.diag(torch.tensor([1.0, 2.0], dtype=torch.double)))
    assert torch.isinf(d.u_min) and torch.isinf(d.u_max)                           # :124-126


def test_zero_dim_sigma_and_bounds():
    """reference tests/pendulum.py:25,76-77 and test_mppi.py:276-291"""
    m = models.Pendulum()
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.tensor(10.0, dtype=torch.double), num_samples=100, horizon=15,
                u_min=torch.tensor(-2.0, dtype=torch.double), u_max=torch.tensor(2.0, dtype=torch.double))
    assert c.nu == 1 and c.noise_sigma.shape == (1, 1) and c.U.shape == (15, 1)
    assert c._vec(c.u_min).shape == (1,)


def test_get_params_format_and_horizon_shift():
    m = _lin()
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), num_samples=100, horizon=10)
    s = c.get_params()
    assert "K=100" in s and "T=10" in s and "lambda=1.0" in s                      # test_mppi.py:324-328
    U0 = c.U.clone()
    c.u_init = torch.tensor([0.5, -0.5], dtype=torch.double)
    c.shift_nominal_trajectory()                                                   # test_mppi.py:293-303
    assert torch.equal(c.U[:-1], U0[1:]) and torch.equal(c.U[-1], c.u_init)
    c.change_horizon(5)
    assert c.T == 5 and c.U.shape == (5, 2)
    c.change_horizon(8)
    assert c.U.shape == (8, 2) and torch.equal(c.U[5:], c.u_init.repeat(3, 1))
    c.reset()
    assert c.U.shape == (8, 2)


def test_generator_consumption_matches_reference():
    """mppi.py:144-145: construction without U_init consumes exactly randn(T,nu)."""
    m = _lin()
    torch.manual_seed(3)
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), horizon=7)
    after = torch.randn(1)
    torch.manual_seed(3)
    ref = torch.randn(7, 2, dtype=torch.double)
    after_ref = torch.randn(1)
    assert torch.equal(c.U, ref) and torch.equal(after, after_ref)


def test_native_model_detection():
    m = _lin()
    f = lambda s, a: s + a
    assert models.native_model_of(m.dynamics, m.running_cost) is m
    assert models.native_model_of(m, m.running_cost) is m
    assert models.native_model_of(m.dynamics, m.running_cost, m.terminal_state_cost) is m
    assert models.native_model_of(f, m.running_cost) is None
    assert models.native_model_of(m.dynamics, lambda s, a: s.sum(-1)) is None
    assert models.native_model_of(m.dynamics, _lin().running_cost) is None          # different object
    assert models.native_model_of(models.Pendulum().dynamics, models.Pendulum().running_cost, lambda s, a: 0) is None
    # the built-in models are time-invariant and take (state, action, t) too: fused under either setting (mppi.py:147-154)
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), step_dependent_dynamics=True)
    assert c._model is m
    x, u = torch.zeros(3, 2, dtype=torch.double), torch.ones(3, 2, dtype=torch.double)
    assert torch.equal(m.dynamics(x, u, 4), m.dynamics(x, u)) and torch.equal(m.running_cost(x, u, 4), m.running_cost(x, u))


def test_native_models_equal_oracle_callables():
    from oracle import dynamics as dyn
    g = torch.Generator().manual_seed(0)
    x = torch.randn(50, 2, generator=g, dtype=torch.double) * 3
    u = torch.randn(50, 1, generator=g, dtype=torch.double) * 3
    P = models.Pendulum()
    assert torch.allclose(P.dynamics(x, u), dyn.pendulum_dynamics(x, u), rtol=0, atol=1e-15)
    assert torch.allclose(P.running_cost(P.dynamics(x, u), u), dyn.pendulum_cost(dyn.pendulum_dynamics(x, u), u), rtol=0, atol=1e-15)
    x6 = torch.randn(20, 6, generator=g)
    u4 = torch.randn(20, 4, generator=g)
    f, q = dyn.make_quadtoy(6, 4)
    I = models.Integrator(6, 4)
    assert torch.equal(I.dynamics(x6, u4), f(x6, u4)) and torch.equal(I.running_cost(x6, u4), q(x6, u4))
    W = dyn.make_mlp_weights(16, 4, 32, seed=2)
    M = models.MLPResidual(*W, 16, 4)
    fm, qm = dyn.make_mlp(*W)
    x16, u4b = torch.randn(9, 16, generator=g), torch.randn(9, 4, generator=g)
    assert torch.equal(M.dynamics(x16, u4b), fm(x16, u4b))
    M2 = models.MLPResidual.random(16, 4, 32, seed=2)
    assert torch.equal(M2.W1, W[0]) and torch.equal(M2.b2, W[3])
    blob = M.param_blob("cpu", torch.float32)
    # W1 | b1 | W2 | b2 | res_scale | qx | qu, the hidden axis zero-padded to the matrix-core kernels' next width (32 units -> 64)
    assert (M.hidden_units, M.hidden) == (32, 64) and blob.numel() == 64 * 20 + 64 + 16 * 64 + 16 + 1 + 16 + 4
    assert torch.equal(blob[:32 * 20].reshape(32, 20), W[0]) and not blob[32 * 20:64 * 20].any()
    assert models.MLPResidual(*dyn.make_mlp_weights(6, 2, 50, seed=2), 6, 2).hidden == 50      # (no matrix-core kernel of that shape: as is)
    assert blob[-20:-4].tolist() == [1.0] * 16 and blob[-4:].tolist() == [0.0] * 4       # the plain sum x^2
    qs, qc = torch.linspace(0.5, 2.0, 16), torch.tensor([0.1, 0.2, 0.3, 0.4])
    Mq = models.MLPResidual(*W, 16, 4, q_state=qs, q_control=qc)
    _, qq = dyn.make_mlp(*W, q_state=qs, q_control=qc)
    assert torch.allclose(Mq.running_cost(x16, u4b), qq(x16, u4b), rtol=1e-6) and torch.equal(M.running_cost(x16, u4b), qm(x16, u4b))
    assert torch.allclose(Mq.running_cost(x16, u4b), (qs * x16 ** 2).sum(-1) + (qc * u4b ** 2).sum(-1), rtol=1e-6)


def test_cpu_device_has_no_compute_path():
    m = _lin()
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double))
    with pytest.raises(RuntimeError, match="MI355X"):
        c.command(torch.zeros(2, dtype=torch.double))


def test_unbuilt_features_fail_loudly():
    m = _lin()
    c = pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), rollout_samples=3, device="cpu")
    assert c._model is not None and c._fused_multi_ok()  # M <= 4 copies of the state run inside K1
    c9 = pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), rollout_samples=9, device="cpu")
    assert c9._needs_generic()                           # more than that: the reference's callback loop
    with pytest.raises(RuntimeError):
        c.command(torch.zeros(2, dtype=torch.double))
    with pytest.raises(ValueError):
        pm.MPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), rng="mt19937")


def test_kmppi_operators_and_rbf():
    """constant-W form == the reference's per-sample solve; RBF KAT test_mppi.py:560-570"""
    m = _lin()
    c = pm.KMPPI(m.dynamics, m.running_cost, 2, torch.eye(2, dtype=torch.double), num_samples=10, horizon=10,
                 U_init=torch.zeros(10, 2, dtype=torch.double))
    assert c.num_support_pts == 5 and c.theta.shape == (5, 2)
    W, Wsh, Tk, Hs = orc.kmppi_matrices(10, 5, torch.double)
    assert torch.allclose(c._W, W, atol=1e-12) and torch.allclose(c._W_shift, Wsh, atol=1e-12)
    assert torch.equal(c.Tk[0], Tk) and torch.equal(c.Hs[0], Hs) and c.Tk.shape == (10, 5)
    theta = torch.randn(5, 2, dtype=torch.double)
    traj, _ = c.deparameterize_to_trajectory_single(theta)
    assert torch.allclose(traj, W @ theta, atol=1e-10)
    tb, _ = c.deparameterize_to_trajectory_batch(theta.expand(10, -1, -1).contiguous())
    assert torch.allclose(tb[3], traj, atol=1e-10)
    k = pm.RBFKernel(sigma=1.0)
    tt = torch.tensor([[0.0], [1.0]], dtype=torch.double)
    kk = k(tt, tt)
    assert torch.allclose(kk.diag(), torch.ones(2, dtype=torch.double), atol=1e-6)
    assert abs(float(kk[0, 1]) - np.exp(-0.5)) < 1e-6
    assert "num_support_pts=5" in c.get_params() and "RBFKernel(sigma=1)" in c.get_params()
    c.theta = theta.clone()
    c.shift_nominal_trajectory()
    assert torch.allclose(c.theta, Wsh @ theta, atol=1e-12)
    c.change_horizon(12)                 # the reference leaves Tk/Hs stale here (SURVEY A-15)
    assert c._W.shape == (12, 5)
    c.reset()
    assert float(c.theta.abs().sum()) == 0.0


def test_sampler_slice_bookkeeping():
    s = pm.SpecificActionSampler()
    s.register_sample_start_end(1, 4)
    assert (s.start_idx, s.end_idx, s.slice) == (1, 4, slice(1, 4))
    x = torch.zeros(3)
    assert s.specific_dynamics(x, x, x, 0) is x


@pytest.mark.parametrize("K,G", [(65536, 8), (100, 3), (7, 7), (524288, 8)])
def test_shard_plan_partitions_contiguously(K, G):
    lo_prev = 0
    for r in range(G):
        sp = ShardPlan(K, r, G)
        assert sp.k_offset == lo_prev and sp.K_local >= 1
        assert sp.bounds(r) == (sp.k_offset, sp.k_offset + sp.K_local)
        lo_prev += sp.K_local
    assert lo_prev == K
    with pytest.raises(ValueError):
        ShardPlan(2, 0, 3)


def test_combine_records_equals_global_softmax():
    """the shard records {beta_g, eta_g, P_g} combined in rank order == one global softmax update"""
    g = torch.Generator().manual_seed(1)
    K, T, nu, lam = 200, 6, 3, 4.0
    cost = torch.randn(K, generator=g, dtype=torch.double) * 10 + 50
    noise = torch.randn(K, T, nu, generator=g, dtype=torch.double)
    U = torch.randn(T, nu, generator=g, dtype=torch.double)
    omega, w, beta, eta = orc.weights(cost, lam)
    U_ref = U + torch.einsum("k,ktn->tn", omega, noise)
    recs = []
    for r in range(3):
        lo, hi = ShardPlan(K, r, 3).bounds(r)
        b = cost[lo:hi].min()
        wg = torch.exp(-(1 / lam) * (cost[lo:hi] - b))
        recs.append(torch.cat([b.view(1), wg.sum().view(1), torch.einsum("k,ktn->tn", wg, noise[lo:hi]).reshape(-1)]))
    U_new, beta2, eta2 = combine_records_host(torch.stack(recs), U, lam)
    assert torch.allclose(U_new, U_ref, rtol=1e-12, atol=1e-12)
    assert float(beta2) == float(beta) and abs(float(eta2) - float(eta)) < 1e-9 * float(eta)


def test_philox_known_answers_and_moments():
    """Random123 KATs for Philox4x32-10 + N(0,1) moments + layout map."""
    r = oph.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(x[0]) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = oph.philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(x[0]) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = oph.philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(x[0]) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # Philox4x32-7 (rng="philox7"): Random123's own known-answer vectors for philox4x32 with 7 rounds (kat_vectors)
    r = oph.philox4x32_10([0], [0], [0], [0], 0, 0, rounds=7)
    assert [int(x[0]) for x in r] == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    r = oph.philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff, rounds=7)
    assert [int(x[0]) for x in r] == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    r = oph.philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0, rounds=7)
    assert [int(x[0]) for x in r] == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    z7 = oph.normals_ktn(seed=42, call=1, K=4096, T=16, nu=3, rounds=7)
    assert abs(z7.mean()) < 0.01 and abs(z7.std() - 1) < 0.01 and abs(np.mean(z7 ** 4) - 3.0) < 0.1
    z = oph.normals_ktn(seed=42, call=1, K=4096, T=16, nu=3)
    assert not np.array_equal(z, z7)
    assert z.shape == (4096, 16, 3)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.mean(z ** 4) - 3.0) < 0.1
    # shard independence: rows of a shard == the same global rows
    z2 = oph.normals_ktn(seed=42, call=1, K=100, T=16, nu=3, k_offset=1000)
    assert np.array_equal(z2, z[1000:1100])
    assert not np.array_equal(oph.normals_ktn(42, 2, 8, 16, 3), z[:8])


def test_background_jit_state_machine(monkeypatch):
    """auto_jit="async": the callables are traced at construction, hipcc runs in a background thread, commands stay on the
    callbacks until it has finished, then the controller adopts the fused model (or keeps the callbacks after a failure).
    The compile itself is replaced by a stub here (no hipcc, no GPU)."""
    import threading
    import time
    from pytorch_mppi_amd import jit
    f = lambda s, a: s + 0.1 * a
    q = lambda s, a: (s ** 2).sum(-1)
    c = pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.double), num_samples=8, horizon=3, auto_jit=False)
    assert c._model is None and c._jit_pending is None
    gate = threading.Event()

    class Dummy:
        nx = nu = 2
        traced_ops, name, process_noise = 7, "traced_dummy", None

    def slow_compile(code, *a, **k):
        assert "x[0]" in code["step"]
        gate.wait(5.0)
        return Dummy()
    monkeypatch.setattr(jit, "compile_traced", slow_compile)
    monkeypatch.setattr(jit, "traced_is_cached", lambda *a, **k: False)
    assert c._try_trace(f, q, None, False, background=True) is None
    assert c._jit_pending is not None and c.jit_note.startswith("generic path for now")
    assert c._needs_generic() and c._model is None             # still compiling: callbacks
    c._adopt_background_model()                                # (what command() does first: nothing to adopt yet)
    assert c._model is None and c._jit_pending is not None
    gate.set()
    assert c.wait_for_jit(5.0) and isinstance(c._model, Dummy) and c._jit_pending is None
    assert c.jit_note.startswith("fused") and "background" in c.jit_note
    # a cached object is adopted at once, also in the background mode
    monkeypatch.setattr(jit, "traced_is_cached", lambda *a, **k: True)
    c2 = pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.double), num_samples=8, horizon=3, auto_jit=False)
    assert isinstance(c2._try_trace(f, q, None, False, background=True), Dummy) and c2._jit_pending is None
    # a failed hipcc run leaves the controller on the callbacks, with the reason in jit_note
    monkeypatch.setattr(jit, "traced_is_cached", lambda *a, **k: False)
    monkeypatch.setattr(jit, "compile_traced", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hipcc failed")))
    c3 = pm.MPPI(f, q, 2, torch.eye(2, dtype=torch.double), num_samples=8, horizon=3, auto_jit=False)
    assert c3._try_trace(f, q, None, False, background=True) is None
    assert not c3.wait_for_jit(5.0) and c3._model is None and "hipcc failed" in c3.jit_note


def test_traced_model_follows_its_trainable_tensors_without_recompiling():
    """jit.from_torch on the learned-pendulum fixture (its object is built by __graft_entry__.build(): a cache hit here): the
    network's parameters are the model's parameter vector; optimizer steps, load_state_dict and a storage swap are picked up
    by refresh_params() -- version counters and data pointers -- and bump the model's parameter version (new device blob)."""
    import jit_fixtures as jf
    from pytorch_mppi_amd import jit
    f, q, net = jf.approx_pendulum_callables()
    m = jit.from_torch(f, q, 2, 1)
    assert m.heavy and m._n_params == 1250 and not m._captured and not m.stale()
    flat = lambda: torch.cat([p_.detach().reshape(-1) for p_ in net.parameters()])
    # the order of the vector is the order in which the callables first read the tensors
    assert sorted(m.params.tolist()) == sorted(flat().tolist())
    v0 = m._param_version
    assert not m.refresh_params() and m._param_version == v0
    jf.train_a_little(net, steps=2)
    assert m.refresh_params() and m._param_version == v0 + 1
    assert sorted(m.params.tolist()) == sorted(flat().tolist())
    assert not m.refresh_params()
    net.load_state_dict({k: v * 0.5 for k, v in net.state_dict().items()})
    assert m.refresh_params() and m._param_version == v0 + 2
    for p_ in net.parameters():                       # a storage swap without a version bump (what module.to() does)
        p_.data = p_.data.clone() + 1.0
    assert m.refresh_params() and sorted(m.params.tolist()) == sorted(flat().tolist())
    blob = m.param_blob("cpu", torch.float32)
    assert blob.dtype == torch.float32 and blob.numel() == 1250
