"""Build-container only: pins oracle/mppi_oracle.py bit-for-bit against the LIVE reference
(/root/reference/src, imported through oracle/ref_loader.py with injected z), and re-runs the
reference's own tests of the stubbed dependency (tests/test_batch_wrapper.py:19-47)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import ref_loader

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_loader.reference_available(), reason="no /root/reference here")]


@pytest.mark.parametrize("name", ["pendulum_c1_f64", "linear_full_f64", "linear_sampler_f64", "quadtoy_f32",
                                  "linear_multi_f64", "linear_multi_f32"])
def test_oracle_bitwise_vs_live_reference(name):
    mod, proxy = ref_loader.load_reference()
    cfg, d = gu.load(name)
    dtype = gu.TDT[cfg["dtype"]]
    f, q, term = gu.torch_callables(cfg, d, dtype)
    kw = gu.ctor_tensors(cfg, dtype)
    if cfg["terminal"]:
        kw["terminal_state_cost"] = term
    sampler_actions = gu.t(d, "sampler_actions", dtype) if cfg["sampler_rows"] else None
    if sampler_actions is not None:
        class _S(mod.SpecificActionSampler):
            def sample_trajectories(self, state, info):
                return sampler_actions.clone()
        kw["specific_action_sampler"] = _S()
    ctrl = mod.MPPI(f, q, cfg["nx"], torch.tensor(cfg["sigma"], dtype=dtype), num_samples=cfg["K"],
                    horizon=cfg["T"], device="cpu", U_init=gu.t(d, "U_init", dtype).clone(), **kw)
    outs = gu.oracle_run(cfg, d)
    state = gu.t(d, "state", dtype)
    for s, r in enumerate(outs):
        proxy.queue.append(gu.t(d, f"z{s}", dtype))
        act = ctrl.command(state, shift_nominal_trajectory=bool(d[f"shift{s}"]))
        assert torch.equal(act, r["action"])
        assert torch.equal(ctrl.U, r["U"])
        assert torch.equal(ctrl.cost_total, r["cost_total"])
        assert torch.equal(ctrl.omega, r["omega"])
        assert torch.equal(ctrl.noise, r["noise"])


def test_batch_wrapper_stub_2d_3d():
    ref_loader.load_reference()                 # (installs the stub of the absent dependency: not only behind another test of this file)
    from arm_pytorch_utilities import handle_batch_input

    @handle_batch_input(n=2)
    def add_2d(a, b):
        assert a.ndim == 2 and b.ndim == 2
        return a + b

    @handle_batch_input(n=3)
    def add_3d(a, b):
        assert a.ndim == 3 and b.ndim == 3
        return a + b

    a2 = torch.tensor([[0.1, 0.2, 0.3]]); b2 = torch.tensor([[0.5, -0.2, 0.3]])
    e2 = torch.tensor([[0.6, 0.0, 0.6]])
    a3, b3 = a2[None], b2[None]
    a4, b4 = torch.tile(a3, [2, 1, 1, 1]), torch.tile(b3, [2, 1, 1, 1])
    assert torch.allclose(add_2d(a2, b2), e2)
    assert torch.allclose(add_2d(a3, b3), e2[None])
    assert torch.allclose(add_2d(a4, b4), torch.tile(e2[None, None], [2, 1, 1, 1]))
    assert torch.allclose(add_3d(a3, b3), e2[None])
    a4b, b4b = torch.tile(a3, [2, 1, 1]), torch.tile(b3, [2, 1, 1])
    assert torch.allclose(add_3d(a4b, b4b), torch.tile(e2[None], [2, 1, 1]))


def _random_spec(seed):
    """one random configuration of the path's constructor surface (mppi.py:45-184, :445-483, :593-615): controller class, model,
    dtype, sizes, Sigma (0-dim / diagonal / full), every optional keyword with probability ~1/2"""
    r = np.random.RandomState(1000 + seed)
    coin = lambda p=0.5: bool(r.rand() < p)
    kind = ["mppi", "smppi", "kmppi"][seed % 3]
    model = ["pendulum", "linear_goal", "quadtoy", "linear_multi"][(seed // 3) % 4]
    spec = dict(model=model, model_args={}, dtype="f64" if coin(0.7) else "f32", K=int(r.randint(6, 48)), T=int(r.randint(3, 11)),
                steps=3, seed=seed, lambda_=float(np.round(10 ** r.uniform(-0.5, 1.2), 3)))
    if model == "pendulum":
        nx, nu = 2, 1
        spec["sigma"] = float(np.round(r.uniform(0.5, 8.0), 2))
        if coin():
            spec["u_min"], spec["u_max"] = -2.0, 2.0
    else:
        if model == "quadtoy":
            nu = int(r.randint(2, 5))
            nx = nu + int(r.randint(0, 4))
        else:
            nx = nu = 2
            B = np.round(r.uniform(-1, 1, (2, 2)) + np.eye(2), 2)
            spec["model_args"] = dict(B=B.tolist(), goal=np.round(r.uniform(-2, 2, 2), 2).tolist())
        d = np.round(r.uniform(0.3, 2.0, nu), 2)
        S = np.diag(d)
        if coin():                                     # full Sigma: SPD by construction
            L = np.tril(np.round(r.uniform(-0.4, 0.4, (nu, nu)), 2), -1) + np.diag(np.sqrt(d))
            S = np.round(L @ L.T, 4)
            S = (S + S.T) / 2
        spec["sigma"] = S.tolist()
        if coin():
            spec["noise_mu"] = np.round(r.uniform(-0.3, 0.3, nu), 2).tolist()
        if coin():
            spec["u_init"] = np.round(r.uniform(-0.1, 0.1, nu), 2).tolist()
        b = coin(0.6), coin(0.3)
        if b[0]:
            spec["u_max"] = np.round(r.uniform(0.5, 2.0, nu), 2).tolist()        # one-sided: mirrored (mppi.py:112-119)
            if b[1]:
                spec["u_min"] = (-np.round(r.uniform(0.5, 2.0, nu), 2)).tolist()
    spec["nx"], spec["nu"] = nx, nu
    if coin():
        spec["u_scale"] = float(np.round(r.uniform(0.3, 1.5), 2))
    if coin(0.4):
        spec["u_per_command"] = int(r.randint(1, min(4, spec["T"])))
    if coin():
        spec["sample_null_action"] = True
    if coin(0.4):
        spec["noise_abs_cost"] = True
    if model in ("linear_goal", "linear_multi") and coin():
        spec["terminal"] = True
    if coin(0.4) and model != "linear_multi":
        spec["per_sample_state"] = True
    if coin(0.4):
        spec["sampler_rows"] = int(r.randint(1, 4))
    if model == "linear_multi":
        spec["model_args"]["w_scale"] = float(np.round(r.uniform(0.05, 0.3), 2))
        spec.update(rollout_samples=int(r.randint(2, 5)), rollout_var_cost=float(np.round(r.uniform(0, 0.5), 2)),
                    rollout_var_discount=float(np.round(r.uniform(0.8, 1.0), 2)), step_dependent_dynamics=True)
    if kind == "smppi":
        sm = dict(w_action_seq_cost=float(np.round(r.uniform(0.0, 2.0), 2)), delta_t=float(np.round(r.uniform(0.2, 1.0), 2)))
        if coin() and nu > 1:
            sm["action_max"] = np.round(r.uniform(0.5, 1.5, nu), 2).tolist()
        spec["smppi"] = sm
    if kind == "kmppi":
        spec["kmppi"] = True
        if coin():
            spec["S"] = int(r.randint(2, spec["T"] + 1))
    return kind, spec


@pytest.mark.parametrize("seed", range(120))
def test_oracle_vs_live_reference_on_random_configurations(seed):
    """differential test over the constructor surface: a random configuration is run on the LIVE reference with injected draws
    (oracle/gen_golden.py build_case -- the code that wrote the committed fixtures) and replayed through the oracle; three commands,
    every public result.  MPPI / SMPPI: bit for bit; KMPPI: the oracle's constant interpolation matrix against the reference's
    vmap(solve), 1e-9 / 1e-4."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import gen_golden
    kind, spec = _random_spec(seed)
    cfg, d = gen_golden.build_case(f"random{seed}", **spec)
    outs = gu.oracle_run(cfg, d)
    keys = ["action", "U", "cost_total", "omega", "noise", "perturbed_action"] + (["theta", "noise_theta"] if kind == "kmppi" else []) \
        + (["action_sequence"] if kind == "smppi" else [])
    for s, r in enumerate(outs):
        for k in keys:
            ref, got = np.array(d[f"{k}{s}"]), r[k].numpy()
            if kind == "kmppi":
                rtol = 1e-9 if cfg["dtype"] == "f64" else 1e-4
                np.testing.assert_allclose(got, ref, rtol=rtol, atol=rtol * max(1.0, float(np.abs(ref).max())), err_msg=f"{spec} step {s} {k}")
            else:
                assert np.array_equal(got, ref), (spec, s, k, float(np.abs(got - ref).max()))
        if cfg["sampler_rows"]:
            assert tuple(d[f"slice{s}"]) == tuple(r["sampler_slice"])


@pytest.mark.parametrize("seed", range(24))
def test_oracle_vs_live_reference_batched_on_random_configurations(seed):
    """MPPI_Batched (mppi.py:691-873) == N independent oracle commands sharing one draw: random N, K, T, Sigma and keywords on the
    live reference (gen_golden.build_batched_case), action and U of three commands"""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import gen_golden
    r = np.random.RandomState(5000 + seed)
    coin = lambda p=0.5: bool(r.rand() < p)
    d_ = np.round(r.uniform(0.3, 2.0, 2), 2)
    S = np.diag(d_)
    if coin():
        L = np.array([[np.sqrt(d_[0]), 0.0], [np.round(r.uniform(-0.4, 0.4), 2), np.sqrt(d_[1])]])
        S = np.round(L @ L.T, 4)
        S = (S + S.T) / 2
    spec = dict(N=int(r.randint(1, 6)), K=int(r.randint(6, 48)), T=int(r.randint(3, 11)), dtype="f64" if coin(0.7) else "f32", sigma=S.tolist(),
                steps=3, seed=seed, lambda_=float(np.round(10 ** r.uniform(-0.5, 1.2), 3)))
    if coin():
        spec["noise_mu"] = np.round(r.uniform(-0.3, 0.3, 2), 2).tolist()
    if coin(0.6):
        spec["u_max"] = np.round(r.uniform(0.5, 2.0, 2), 2).tolist()
    if coin():
        spec["u_scale"] = float(np.round(r.uniform(0.3, 1.5), 2))
    if coin(0.4):
        spec["u_per_command"] = int(r.randint(1, min(4, spec["T"])))
    if coin(0.4):
        spec["noise_abs_cost"] = True
    cfg, d = gen_golden.build_batched_case(f"random_batched{seed}", **spec)
    outs = gu.oracle_run_batched(cfg, d)
    rtol = 1e-12 if cfg["dtype"] == "f64" else 2e-6
    for s, o in enumerate(outs):
        for k in ("action", "U"):
            ref = np.array(d[f"{k}{s}"])
            np.testing.assert_allclose(o[k].numpy(), ref, rtol=rtol, atol=rtol * max(1.0, float(np.abs(ref).max())), err_msg=f"{spec} step {s} {k}")


@pytest.mark.parametrize("name", ["MPPI", "SMPPI", "KMPPI", "MPPI_Batched"])
def test_constructor_signature_and_public_members_are_the_references(name):
    """the drop-in boundary, checked against the LIVE classes: every constructor parameter of the reference in the same
    position, of the same kind, with the same default; whatever this package adds is keyword-only (rng, seed, shard, auto_jit, devices);
    every public method / property of the reference class exists here"""
    import inspect
    import pytorch_mppi_amd as pm
    mod, _ = ref_loader.load_reference()
    R, O = getattr(mod, name), getattr(pm, name)
    rp = list(inspect.signature(R.__init__).parameters.values())
    op = list(inspect.signature(O.__init__).parameters.values())
    for a, b in zip(rp, op):
        assert (a.name, a.kind) == (b.name, b.kind), (a, b)
        if a.default is inspect.Parameter.empty or b.default is inspect.Parameter.empty:
            assert a.default is b.default, (a, b)
        else:
            assert type(a.default).__name__ == type(b.default).__name__ and str(a.default) == str(b.default), (a, b)
    assert len(op) >= len(rp)
    for b in op[len(rp):]:
        assert b.kind in (inspect.Parameter.KEYWORD_ONLY, inspect.Parameter.VAR_KEYWORD) and b.name in ("rng", "seed", "shard", "auto_jit", "devices", "kwargs"), b
    public = lambda c: {n for n, _ in inspect.getmembers(c) if not n.startswith("_")}
    assert not public(R) - public(O), public(R) - public(O)
    for n in public(R):
        sr, so = getattr(R, n), getattr(O, n)
        if inspect.isfunction(sr):
            pr = [(p.name, p.kind) for p in inspect.signature(sr).parameters.values()]
            po = [(p.name, p.kind) for p in inspect.signature(so).parameters.values()]
            assert po[:len(pr)] == pr, (n, pr, po)
