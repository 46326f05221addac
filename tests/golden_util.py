"""Helpers shared by the CPU and GPU parity tests: load a tests/golden/*.npz fixture (produced
by oracle/gen_golden.py from the LIVE reference) and rebuild the oracle Problem for it."""
import glob
import json
import os

import numpy as np
import torch

from oracle import dynamics as dyn
from oracle import mppi_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TDT = {"f32": torch.float32, "f64": torch.float64}


def golden_names(batched=False, fused=None):
    """Fixture names.  fused=True: only those with a native (fused-kernel) model; the M > 1 fixtures
    (`*_multi_*`, deterministic stand-in for stochastic callbacks) exist on the callback path only."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    names = [n for n in names if n.startswith("batched_") == batched]
    if fused:
        names = [n for n in names if "_multi_" not in n]
    return names


def oracle_run_batched(cfg, d, dtype=None):
    """MPPI_Batched == N independent MPPI commands that share one z (mppi.py:838-869)."""
    dtype = dtype or TDT[cfg["dtype"]]
    f, q, _ = dyn.make_linear_goal(t(d, "B", dtype), t(d, "goal", dtype))
    p = orc.Problem(dynamics=f, running_cost=q, nx=2, noise_sigma=torch.tensor(cfg["sigma"], dtype=dtype),
                    K=cfg["K"], T=cfg["T"], **ctor_tensors(cfg, dtype))
    U = t(d, "U_init", dtype)
    states = t(d, "state", dtype)
    outs = []
    for s in range(cfg["steps"]):
        z = t(d, f"z{s}", dtype)
        rs = [orc.command(p, U[n], states[n], z, bool(d[f"shift{s}"])) for n in range(cfg["N"])]
        U = torch.stack([r["U"] for r in rs])
        outs.append(dict(U=U, action=torch.stack([r["action"] for r in rs]),
                         cost_total=torch.stack([r["cost_total"] for r in rs]),
                         omega=torch.stack([r["omega"] for r in rs])))
    return outs


def load(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(str(d["config"]))
    return cfg, d


def t(d, key, dtype):
    return torch.from_numpy(np.array(d[key])).to(dtype)


def ctor_tensors(cfg, dtype):
    kw = {}
    for k, v in cfg["ctor"].items():
        if k in ("u_min", "u_max", "noise_mu", "u_init") or isinstance(v, list):
            kw[k] = torch.tensor(v, dtype=dtype)
        else:
            kw[k] = v
    return kw


def torch_callables(cfg, d, dtype, device="cpu"):
    """(dynamics, running_cost, terminal) torch callables of the fixture's model."""
    m = cfg["model"]
    if m == "linear_multi":
        tt = lambda key: t(d, key, dtype).to(device)
        return dyn.make_linear_goal_multi(tt("B"), tt("goal"), tt("w"), cfg["K"])
    if device != "cpu":
        tt = lambda key: t(d, key, dtype).to(device)
        if m == "linear_goal":
            return dyn.make_linear_goal(tt("B"), tt("goal"))
        if m == "mlp":
            f, q = dyn.make_mlp(tt("W1"), tt("b1"), tt("W2"), tt("b2"), cfg["model_args"].get("res_scale", 0.1))
            return f, q, None
    if m == "pendulum":
        return dyn.pendulum_dynamics, dyn.pendulum_cost, None
    if m == "quadtoy":
        f, q = dyn.make_quadtoy(cfg["nx"], cfg["nu"])
        return f, q, None
    if m == "linear_goal":
        return dyn.make_linear_goal(t(d, "B", dtype), t(d, "goal", dtype))
    if m == "mlp":
        f, q = dyn.make_mlp(t(d, "W1", dtype), t(d, "b1", dtype), t(d, "W2", dtype), t(d, "b2", dtype),
                            cfg["model_args"].get("res_scale", 0.1))
        return f, q, None
    raise ValueError(m)


def oracle_problem(cfg, d, dtype=None):
    dtype = dtype or TDT[cfg["dtype"]]
    f, q, term = torch_callables(cfg, d, dtype)
    kw = ctor_tensors(cfg, dtype)
    return orc.Problem(dynamics=f, running_cost=q, nx=cfg["nx"],
                       noise_sigma=torch.tensor(cfg["sigma"], dtype=dtype), K=cfg["K"], T=cfg["T"],
                       terminal_state_cost=term if cfg["terminal"] else None, **kw)


def oracle_run(cfg, d, dtype=None):
    """Replays the fixture through the oracle; returns the list of per-step result dicts."""
    dtype = dtype or TDT[cfg["dtype"]]
    p = oracle_problem(cfg, d, dtype)
    U = t(d, "U_init", dtype)
    state = t(d, "state", dtype)
    sampler = t(d, "sampler_actions", dtype) if cfg["sampler_rows"] else None
    outs = []
    if cfg.get("smppi"):
        sm = dict(cfg["smppi"])
        amax = torch.tensor(sm["action_max"], dtype=dtype) if "action_max" in sm else torch.tensor(float("inf"))
        amin = -amax
        A = U.clone()                       # U_init = initial action sequence (mppi.py:479-483)
        U = torch.zeros_like(U)
        for s in range(cfg["steps"]):
            r = orc.smppi_command(p, U, A, state, t(d, f"z{s}", dtype), amin, amax, sm.get("w_action_seq_cost", 1.0),
                                  sm.get("delta_t", 1.0), bool(d[f"shift{s}"]), sampler)
            U, A = r["U"], r["action_sequence"]
            outs.append(r)
        return outs
    if cfg["kmppi"]:
        W, W_shift, _, _ = orc.kmppi_matrices(cfg["T"], cfg["S"], dtype)
        theta = torch.zeros(cfg["S"], cfg["nu"], dtype=dtype)
    for s in range(cfg["steps"]):
        z = t(d, f"z{s}", dtype)
        shift = bool(d[f"shift{s}"])
        if cfg["kmppi"]:
            r = orc.kmppi_command(p, theta, U, state, z, W, W_shift, shift, sampler)
            theta = r["theta"]
        else:
            r = orc.command(p, U, state, z, shift, sampler)
        U = r["U"]
        outs.append(r)
    return outs


def native_model(cfg, d, dtype):
    """The engine's NativeModel for a fixture (pytorch_mppi_amd.models)."""
    from pytorch_mppi_amd import models
    m = cfg["model"]
    if m == "pendulum":
        return models.Pendulum()
    if m == "quadtoy":
        return models.Integrator(cfg["nx"], cfg["nu"])
    if m == "linear_goal":
        return models.LinearGoal(t(d, "B", dtype), t(d, "goal", dtype))
    if m == "mlp":
        return models.MLPResidual(t(d, "W1", dtype), t(d, "b1", dtype), t(d, "W2", dtype), t(d, "b2", dtype),
                                  cfg["nx"], cfg["nu"], cfg["model_args"].get("res_scale", 0.1))
    raise ValueError(m)


def engine_controller(cfg, d, *, native=True, device="cuda", dtype=None, **extra):
    """Build pytorch_mppi_amd.MPPI / KMPPI for a fixture, on the fused (native model) or the
    generic (plain torch callables) path."""
    import pytorch_mppi_amd as pm
    dtype = dtype or TDT[cfg["dtype"]]
    kw = ctor_tensors(cfg, dtype)
    if native:
        model = native_model(cfg, d, dtype)
        f, q = model.dynamics, model.running_cost
        term = model.terminal_state_cost if cfg["terminal"] else None
    else:
        f, q, term = torch_callables(cfg, d, dtype, device)
        term = term if cfg["terminal"] else None
    if term is not None:
        kw["terminal_state_cost"] = term
    if cfg["sampler_rows"]:
        sa = t(d, "sampler_actions", dtype)

        class _S(pm.SpecificActionSampler):
            def sample_trajectories(self, state, info):
                return sa.clone()

        kw["specific_action_sampler"] = _S()
    kw.update(extra)
    if cfg.get("smppi"):
        for k2, v2 in cfg["smppi"].items():
            kw[k2] = torch.tensor(v2, dtype=dtype) if isinstance(v2, list) else v2
    cls = pm.KMPPI if cfg["kmppi"] else (pm.SMPPI if cfg.get("smppi") else pm.MPPI)
    if cfg["kmppi"]:
        kw["num_support_pts"] = cfg["S"]
    return cls(f, q, cfg["nx"], torch.tensor(cfg["sigma"], dtype=dtype), num_samples=cfg["K"],
               horizon=cfg["T"], device=device, U_init=t(d, "U_init", dtype).clone(), **kw)


def problem_as(p, dtype, **override):
    """The same oracle Problem in another dtype -- the fp32 twin whose distance from the fp64 run is the noise floor of
    the SURVEY 7.3 criterion.  Callables that capture tensors of a fixed dtype must be overridden."""
    import dataclasses
    kw = {}
    for f in dataclasses.fields(orc.Problem):
        if not f.init:
            continue
        v = getattr(p, f.name)
        kw[f.name] = v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v
    kw.update(override)
    return orc.Problem(**kw)
