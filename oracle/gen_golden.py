"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the LIVE reference
(/root/reference/src, v0.9.1) in the build container with injected standard-normal draws.

    python oracle/gen_golden.py            # rewrites every fixture

The reference holds no golden vectors of its own for controller outputs, so these
fixtures are the pinned known-answers: inputs (z, U0, state, parameters) and every public
result of each `command()` call.  They travel to the GPU box; /root/reference does not.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dynamics as dyn                      # noqa: E402
from ref_loader import load_reference       # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _np(x):
    if x is None:
        return np.zeros(0)
    # copy: SMPPI updates `action_sequence` in place (mppi.py:515), a numpy VIEW recorded at step s
    # would silently change at step s+1
    return x.detach().cpu().numpy().copy() if torch.is_tensor(x) else np.array(x)


def run_case(name, **spec):
    """one fixture: build_case() on the live reference, written to tests/golden/<name>.npz"""
    cfg, out = build_case(name, **spec)
    out["config"] = np.array(json.dumps(cfg))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: action0={out['action0']}")


def build_case(name, *, model, model_args, nx, nu, K, T, dtype, sigma, steps=2, state=None,
               per_sample_state=False, kmppi=False, S=None, sampler_rows=0, terminal=False,
               seed=0, smppi=None, **ctor):
    """(config, arrays) of one case run on the LIVE reference with injected draws -- what a fixture file holds; also what
    tests/test_oracle_reference.py's random-configuration test replays through the oracle without writing anything"""
    mod, proxy = load_reference()
    tdt = {"f32": torch.float32, "f64": torch.float64}[dtype]
    g = torch.Generator().manual_seed(seed)
    extra = {}
    if model == "pendulum":
        f, q, term = dyn.pendulum_dynamics, dyn.pendulum_cost, None
    elif model == "quadtoy":
        f, q = dyn.make_quadtoy(nx, nu)
        term = None
    elif model == "linear_goal":
        B = torch.tensor(model_args["B"], dtype=tdt)
        goal = torch.tensor(model_args["goal"], dtype=tdt)
        f, q, term = dyn.make_linear_goal(B, goal)
        extra["B"], extra["goal"] = _np(B), _np(goal)
    elif model == "linear_multi":
        # M > 1 rollouts (mppi.py:334-373): the M copies differ by a fixed disturbance table
        B = torch.tensor(model_args["B"], dtype=tdt)
        goal = torch.tensor(model_args["goal"], dtype=tdt)
        w = torch.randn(ctor["rollout_samples"], T, nx, generator=g, dtype=tdt) * model_args["w_scale"]
        f, q, term = dyn.make_linear_goal_multi(B, goal, w, K)
        extra["B"], extra["goal"], extra["w"] = _np(B), _np(goal), _np(w)
    elif model == "mlp":
        W1, b1, W2, b2 = dyn.make_mlp_weights(nx, nu, model_args["hidden"], seed=2, dtype=tdt)
        f, q = dyn.make_mlp(W1, b1, W2, b2, model_args.get("res_scale", 0.1))
        term = None
        extra.update(W1=_np(W1), b1=_np(b1), W2=_np(W2), b2=_np(b2))
    else:
        raise ValueError(model)
    sigma_t = torch.tensor(sigma, dtype=tdt)
    kw = {}
    for k, v in ctor.items():
        kw[k] = torch.tensor(v, dtype=tdt) if isinstance(v, (list, tuple)) or k in ("u_min", "u_max", "noise_mu", "u_init") else v
    if terminal:
        kw["terminal_state_cost"] = term
    sampler = None
    sampler_actions = None
    if sampler_rows:
        sampler_actions = torch.randn(sampler_rows, T, nu, generator=g, dtype=tdt) * 0.5

        class _S(mod.SpecificActionSampler):
            def sample_trajectories(self, state, info):
                return sampler_actions.clone()

        sampler = _S()
        kw["specific_action_sampler"] = sampler
    U0 = torch.randn(T, nu, generator=g, dtype=tdt) * 0.3
    cls = mod.KMPPI if kmppi else (mod.SMPPI if smppi is not None else mod.MPPI)
    if kmppi and S is not None:
        kw["num_support_pts"] = S
    if smppi is not None:
        for k2, v2 in smppi.items():
            kw[k2] = torch.tensor(v2, dtype=tdt) if isinstance(v2, (list, tuple)) else v2
    ctrl = cls(f, q, nx, sigma_t, num_samples=K, horizon=T, device="cpu", U_init=U0.clone(), **kw)
    if state is None:
        state = torch.randn((K, nx) if per_sample_state else (nx,), generator=g, dtype=tdt)
    else:
        state = torch.tensor(state, dtype=tdt)
    out = dict(U_init=_np(U0), state=_np(state), **extra)
    if sampler_actions is not None:
        out["sampler_actions"] = _np(sampler_actions)
    zshape = (K, ctrl.num_support_pts if kmppi else T, nu)
    for s in range(steps):
        z = torch.randn(*zshape, generator=g, dtype=tdt)
        proxy.queue.append(z)
        shift = (s % 2 == 0) if steps > 2 else True
        act = ctrl.command(state, shift_nominal_trajectory=shift)
        assert not proxy.queue
        out[f"z{s}"] = _np(z)
        out[f"shift{s}"] = np.array(shift)
        out[f"action{s}"] = _np(act)
        out[f"U{s}"] = _np(ctrl.U)
        out[f"cost_total{s}"] = _np(ctrl.cost_total)
        out[f"omega{s}"] = _np(ctrl.omega)
        out[f"noise{s}"] = _np(ctrl.noise)
        out[f"perturbed_action{s}"] = _np(ctrl.perturbed_action)
        if smppi is not None:
            out[f"action_sequence{s}"] = _np(ctrl.action_sequence)
        if kmppi:
            out[f"theta{s}"] = _np(ctrl.theta)
            out[f"noise_theta{s}"] = _np(ctrl.noise_theta)
        if sampler is not None:
            out[f"slice{s}"] = np.array([sampler.start_idx, sampler.end_idx])
    cfg = dict(name=name, model=model, model_args=model_args, nx=nx, nu=nu, K=K, T=T, dtype=dtype,
               sigma=sigma, steps=steps, per_sample_state=per_sample_state, kmppi=kmppi,
               S=(int(ctrl.num_support_pts) if kmppi else None), sampler_rows=sampler_rows,
               terminal=terminal, ctor=ctor, smppi=smppi, reference="UM-ARM-Lab/pytorch_mppi v0.9.1",
               torch=torch.__version__)
    return cfg, out


def run_batched_case(name, **spec):
    cfg, out = build_batched_case(name, **spec)
    out["config"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: action0={out['action0'][0]}")


def build_batched_case(name, *, N, K, T, dtype, sigma, steps=2, seed=0, **ctor):
    """MPPI_Batched (mppi.py:691-873) on the linear-goal environment: U (N,T,nu), shared z per command."""
    mod, proxy = load_reference()
    tdt = {"f32": torch.float32, "f64": torch.float64}[dtype]
    g = torch.Generator().manual_seed(seed)
    B = torch.tensor([[1.0, 0.0], [0.0, -1.0]], dtype=tdt)
    goal = torch.tensor([2.0, 2.0], dtype=tdt)
    f, q, _ = dyn.make_linear_goal(B, goal)
    kw = {k: (torch.tensor(v, dtype=tdt) if isinstance(v, (list, tuple)) else v) for k, v in ctor.items()}
    nu = 2
    U0 = torch.randn(N, T, nu, generator=g, dtype=tdt) * 0.3
    proxy.queue.append(torch.zeros(N, T, nu, dtype=tdt))          # the constructor's own draw (:797), replaced below
    ctrl = mod.MPPI_Batched(f, q, 2, torch.tensor(sigma, dtype=tdt), N, num_samples=K, horizon=T, device="cpu", **kw)
    ctrl.U = U0.clone()
    states = torch.randn(N, 2, generator=g, dtype=tdt) * 2
    out = dict(U_init=_np(U0), state=_np(states), B=_np(B), goal=_np(goal))
    for s in range(steps):
        z = torch.randn(K, T, nu, generator=g, dtype=tdt)
        proxy.queue.append(z)
        shift = (s % 2 == 0)
        act = ctrl.command(states, shift_nominal_trajectory=shift)
        assert not proxy.queue
        out[f"z{s}"], out[f"shift{s}"], out[f"action{s}"], out[f"U{s}"] = _np(z), np.array(shift), _np(act), _np(ctrl.U)
    cfg = dict(name=name, model="linear_goal", nx=2, nu=nu, N=N, K=K, T=T, dtype=dtype, sigma=sigma, steps=steps,
               ctor=ctor, batched=True, reference="UM-ARM-Lab/pytorch_mppi v0.9.1", torch=torch.__version__)
    return cfg, out


def main():
    I2 = [[1.0, 0.0], [0.0, 1.0]]
    Bt = [[1.0, 0.0], [0.0, -1.0]]
    # C1: the reference's own CPU-runnable case, tests/pendulum.py (0-dim sigma, 0-dim bounds)
    run_case("pendulum_c1_f64", model="pendulum", model_args={}, nx=2, nu=1, K=100, T=15, dtype="f64",
             sigma=10.0, state=[3.141592653589793, 1.0], steps=3, lambda_=1.0, u_min=-2.0, u_max=2.0)
    run_case("pendulum_f32", model="pendulum", model_args={}, nx=2, nu=1, K=256, T=32, dtype="f32",
             sigma=10.0, state=[3.141592653589793, 1.0], steps=2, lambda_=1.0, u_min=-2.0, u_max=2.0, seed=1)
    # test_mppi.py environment: diag sigma
    run_case("linear_diag_f64", model="linear_goal", model_args=dict(B=Bt, goal=[2.0, 2.0]), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[-3.0, -2.0], steps=3, lambda_=1.0, seed=2)
    # full sigma + mu + scale + null action + abs cost + one-sided bound + u_per_command
    run_case("linear_full_f64", model="linear_goal", model_args=dict(B=Bt, goal=[2.0, 2.0]), nx=2, nu=2,
             K=128, T=12, dtype="f64", sigma=[[1.0, 0.4], [0.4, 0.5]], state=[-1.0, 0.5], steps=2,
             lambda_=2.5, noise_mu=[0.1, -0.2], u_scale=0.5, sample_null_action=True, noise_abs_cost=True,
             u_max=[1.5, 1.0], u_per_command=3, u_init=[0.05, -0.05], seed=3)
    # terminal cost + sampler rows + null action (global row bookkeeping)
    run_case("linear_sampler_f64", model="linear_goal", model_args=dict(B=Bt, goal=[2.0, 2.0]), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[0.0, 0.0], steps=2, lambda_=1.0,
             sampler_rows=3, sample_null_action=True, terminal=True, seed=4)
    # per-sample initial states, fp32, healthy N_eff
    run_case("quadtoy_f32", model="quadtoy", model_args={}, nx=6, nu=4, K=256, T=12, dtype="f32",
             sigma=[[1, 0, 0, 0], [0, 2, 0, 0], [0, 0, 0.5, 0], [0, 0, 0, 1.5]], steps=2, lambda_=40.0,
             per_sample_state=True, seed=5)
    run_case("quadtoy16_f32", model="quadtoy", model_args={}, nx=16, nu=12, K=128, T=16, dtype="f32",
             sigma=np.eye(12).tolist(), steps=2, lambda_=60.0, seed=6)
    run_case("mlp_f32", model="mlp", model_args=dict(hidden=256), nx=16, nu=4, K=128, T=8, dtype="f32",
             sigma=np.eye(4).tolist(), steps=2, lambda_=3.0, seed=7)
    run_case("mlp_f64", model="mlp", model_args=dict(hidden=256), nx=16, nu=4, K=64, T=8, dtype="f64",
             sigma=np.eye(4).tolist(), steps=1, lambda_=3.0, seed=7)
    # SMPPI (lifted control): action bounds, delta_t, smoothness weight; U_init = initial action sequence
    run_case("smppi_linear_f64", model="linear_goal", model_args=dict(B=Bt, goal=[2.0, 2.0]), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[-3.0, -2.0], steps=3, lambda_=1.0,
             smppi=dict(w_action_seq_cost=0.7, delta_t=0.5, action_max=[1.0, 0.8]), u_max=[2.0, 2.0], seed=10)
    run_case("smppi_quadtoy_f32", model="quadtoy", model_args={}, nx=6, nu=4, K=128, T=12, dtype="f32",
             sigma=[[1, 0.2, 0, 0], [0.2, 2, 0, 0], [0, 0, 0.5, 0], [0, 0, 0, 1.5]], steps=2, lambda_=25.0,
             smppi=dict(w_action_seq_cost=2.0, delta_t=1.0), sample_null_action=True, u_scale=0.5, seed=11)
    # M = 3 rollouts per action sequence + discounted variance cost + terminal cost (mppi.py:334-373)
    run_case("linear_multi_f64", model="linear_multi", model_args=dict(B=Bt, goal=[2.0, 2.0], w_scale=0.15), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[-3.0, -2.0], steps=3, lambda_=1.0, terminal=True,
             rollout_samples=3, rollout_var_cost=0.1, rollout_var_discount=0.9, step_dependent_dynamics=True,
             sample_null_action=True, u_max=[1.5, 1.0], seed=14)
    run_case("linear_multi_f32", model="linear_multi", model_args=dict(B=Bt, goal=[1.0, -1.0], w_scale=0.3), nx=2, nu=2,
             K=256, T=12, dtype="f32", sigma=[[1.0, 0.3], [0.3, 0.6]], steps=2, lambda_=6.0,
             rollout_samples=4, rollout_var_cost=0.5, step_dependent_dynamics=True, per_sample_state=True, seed=15)
    # ... and the same M rollouts under the other two controllers (both call _compute_rollout_costs, mppi.py:564 / :672)
    run_case("smppi_multi_f64", model="linear_multi", model_args=dict(B=Bt, goal=[2.0, 2.0], w_scale=0.15), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[-3.0, -2.0], steps=3, lambda_=1.0,
             rollout_samples=3, rollout_var_cost=0.2, rollout_var_discount=0.9, step_dependent_dynamics=True,
             smppi=dict(w_action_seq_cost=0.7, delta_t=0.5, action_max=[1.0, 0.8]), sample_null_action=True, seed=16)
    run_case("kmppi_multi_f64", model="linear_multi", model_args=dict(B=Bt, goal=[1.0, -1.0], w_scale=0.2), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[-3.0, -2.0], steps=3, lambda_=1.0, kmppi=True,
             rollout_samples=2, rollout_var_cost=0.3, step_dependent_dynamics=True, u_max=[1.0, 1.0], seed=17)
    run_batched_case("batched_linear_f64", N=3, K=100, T=10, dtype="f64", sigma=[[1.0, 0.0], [0.0, 1.0]], steps=3,
                     lambda_=1.0, u_max=[1.5, 1.0], seed=12)
    run_batched_case("batched_linear_full_f32", N=5, K=128, T=8, dtype="f32", sigma=[[1.0, 0.3], [0.3, 0.6]], steps=2,
                     lambda_=4.0, noise_mu=[0.05, -0.1], u_scale=0.5, u_per_command=2, noise_abs_cost=True, seed=13)
    run_case("kmppi_linear_f64", model="linear_goal", model_args=dict(B=Bt, goal=[2.0, 2.0]), nx=2, nu=2,
             K=100, T=10, dtype="f64", sigma=I2, state=[-3.0, -2.0], steps=3, lambda_=1.0, kmppi=True,
             u_max=[1.0, 1.0], seed=8)
    run_case("kmppi_quadtoy_f32", model="quadtoy", model_args={}, nx=6, nu=4, K=128, T=16, dtype="f32",
             sigma=np.eye(4).tolist(), steps=2, lambda_=30.0, kmppi=True, S=6, seed=9)


if __name__ == "__main__":
    main()
