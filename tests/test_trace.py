"""pytorch_mppi_amd/trace.py: plain torch callables -> C++ functor bodies.  CPU tests: the traced
bodies are compiled for the host and compared with the callables on random batches (the same check `jit.from_torch`
runs before it hands the code to hipcc); callables outside the traceable subset are refused, not mistranslated."""
import math

import numpy as np
import pytest
import torch

import jit_fixtures as jf
from pytorch_mppi_amd import trace


def _roundtrip(f, q, nx, nu, term=None, step_dependent=False):
    code = trace.generate(f, q, nx, nu, term, step_dependent)
    assert trace.verify_on_host(code, f, q, nx, nu, term, step_dependent)
    return code


def test_reference_pendulum_with_numpy_ufuncs_on_tensors():
    f, q = jf.ref_pendulum_callables()
    code = _roundtrip(f, q, 2, 1)
    assert "m_sin" in code["step"] and "clampT" in code["step"] and "m_floor" in code["cost"]


def test_reference_linear_dynamics_goal_cost_terminal():
    f, q, t = jf.ref_linear_callables()
    code = _roundtrip(f, q, 2, 2, t)
    assert code["terminal"] is not None and "u[1]" in code["step"]
    assert "* T(0.0)" not in code["step"], "structural zeros of the constant matrix are dropped"


def test_unicycle_stack_and_trig_and_terminal():
    f, q, t = jf.unicycle_callables()
    _roundtrip(f, q, 3, 2, t)


def test_step_dependent_callables_see_the_timestep():
    f, q = jf.drifting_callables()
    code = _roundtrip(f, q, 2, 2, step_dependent=True)
    assert "T(t)" in code["step"] and "T(t)" in code["cost"]


def test_nn_module_dynamics():
    f, q = jf.small_mlp_callables()
    code = _roundtrip(f, q, 4, 2)
    assert code["step"].count("m_tanh") == 8


def test_assorted_elementwise_ops_and_where():
    w = torch.tensor([0.5, -1.5, 2.0], dtype=torch.float64)

    def f(s, a):
        v = torch.where(s[:, :1] > 0.2, torch.sqrt(torch.abs(s[:, :1]) + 1.0), torch.exp(-s[:, 1:2] ** 2))
        r = torch.maximum(s[:, 1:2], a[:, :1]) * torch.sigmoid(a[:, 1:2]) + torch.atan2(s[:, 2:3], 1.0 + s[:, 0:1] ** 2)
        z = (s * w).sum(dim=1, keepdim=True) / 3.0 + torch.tanh(a).mean(dim=1, keepdim=True)
        return torch.cat((v, r, z), dim=1)

    def q(s, a):
        return (s.clamp(min=-1.0) ** 3).sum(-1) + torch.minimum(a[:, 0], a[:, 1]).abs() + torch.log(1.0 + (s ** 2).sum(-1)) + (s[:, 0] % 0.7)

    _roundtrip(f, q, 3, 2)


def test_in_place_write_into_the_state_is_local():
    def f(s, a):
        s = s.clone()
        s[:, 0] = s[:, 0] + a[:, 0]
        s[:, 1] += 0.5 * a[:, 0]
        return s

    _roundtrip(f, lambda s, a: (s ** 2).sum(-1), 2, 1)


def test_trainable_parameters_are_not_baked_in_and_captured_tensors_are_watched():
    net = torch.nn.Linear(3, 2).double()
    f = lambda s, a: s + net(torch.cat((s, a), dim=1))
    q = lambda s, a: (s ** 2).sum(-1)
    with pytest.raises(trace.TraceUnsupported, match="requires grad"):
        trace.generate(f, q, 2, 1)
    for p_ in net.parameters():
        p_.requires_grad_(False)
    code = trace.generate(f, q, 2, 1)
    caps = code["captured"]
    assert len(caps) >= 2 and all(t._version == v for t, v in caps)
    with torch.no_grad():
        net.weight.mul_(2.0)
    assert any(t._version != v for t, v in caps), "an in-place update of a captured tensor is visible to the watcher"


@pytest.mark.parametrize("bad", ["control_flow", "item", "numpy", "shape", "constant", "uses_earlier_state"])
def test_untraceable_callables_are_refused(bad):
    q = lambda s, a: (s ** 2).sum(-1)
    term = None
    if bad == "control_flow":
        def f(s, a):
            return s + a if s.sum() > 0 else s - a
    elif bad == "item":
        def f(s, a):
            return s * float(a[0, 0])
    elif bad == "numpy":
        def f(s, a):
            return torch.as_tensor(np.asarray(s)) + a
    elif bad == "shape":
        def f(s, a):
            return torch.cat((s, a), dim=1)          # augmented state: more than nx columns
    elif bad == "constant":
        def f(s, a):
            return torch.zeros(1, 2, dtype=torch.float64)
    else:
        f = lambda s, a: s + a
        term = lambda states, actions: (states[..., 0, :] ** 2).sum(-1)
    with pytest.raises(trace.TraceUnsupported):
        trace.generate(f, q, 2, 2, term)


def test_verification_catches_a_wrong_translation():
    f, q = jf.ref_pendulum_callables()
    code = trace.generate(f, q, 2, 1)
    code["step"] = code["step"].replace("T(0.05)", "T(0.06)")
    with pytest.raises(trace.TraceUnsupported):
        trace.verify_on_host(code, f, q, 2, 1)
