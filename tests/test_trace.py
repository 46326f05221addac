"""pytorch_mppi_amd/trace.py: plain torch callables -> C++ functor bodies.  CPU tests: the traced
bodies are compiled for the host and compared with the callables on random batches (the same check `jit.from_torch`
runs before it hands the code to hipcc); callables outside the traceable subset are refused, not mistranslated."""
import math

import numpy as np
import pytest
import torch

import jit_fixtures as jf
from pytorch_mppi_amd import jit, trace


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


def test_trainable_parameters_are_read_at_run_time_and_captured_tensors_are_watched():
    """nn.Parameter / requires_grad tensors are elements of the model's parameter vector p[] (their values change under the
    user's training loop: /root/reference/tests/pendulum_approximate.py:140-170); plain tensors become constants and
    their version counters are watched."""
    net = torch.nn.Linear(3, 2).double()
    scale = torch.tensor([2.0, 0.5], dtype=torch.float64)
    f = lambda s, a: s + scale * net(torch.cat((s, a), dim=1))
    q = lambda s, a: (s ** 2).sum(-1)
    code = _roundtrip(f, q, 2, 1)
    assert code["n_params"] == 8 and "p[7]" in code["step"] and [b for _, b in code["param_tensors"]] in ([0, 6], [0, 2])
    caps = code["captured"]
    assert len(caps) == 1 and caps[0][0] is scale and all(t._version == v for t, v in caps)
    # the SAME code follows the parameters: an optimizer-style in-place update, frozen or not
    with torch.no_grad():
        net.weight.mul_(-1.5)
        net.bias.add_(0.25)
    for p_ in net.parameters():
        p_.requires_grad_(False)
    assert trace.verify_on_host(code, f, q, 2, 1)
    with torch.no_grad():
        scale.mul_(2.0)
    assert any(t._version != v for t, v in caps), "an in-place update of a captured tensor is visible to the watcher"
    with pytest.raises(trace.TraceUnsupported):
        trace.verify_on_host(code, f, q, 2, 1)        # ... and the stale constant is what the check catches


def test_learned_pendulum_dynamics_of_the_reference_example():
    """tests/pendulum_approximate.py:47-67 of the reference: 3 -> 32 -> 32 -> 2 tanh residual network, trainable, angle wrapped
    by item assignment.  Traced once; the functor follows the network through training steps."""
    f, q, net = jf.approx_pendulum_callables()
    code = _roundtrip(f, q, 2, 1)
    # (the three Linear layers stay layers -- one chain, the two tanh applied to whole register arrays; csrc/mlp_wide.hpp)
    assert code["n_params"] == 3 * 32 + 32 + 32 * 32 + 32 + 32 * 2 + 2 and code["step"].count("m_tanh") == 2 and len(code["dense"]) == 3
    assert not code["captured"]
    jf.train_a_little(net)
    assert trace.verify_on_host(code, f, q, 2, 1)
    # an fp32 network is checked at fp32 accuracy
    f32, q32, _ = jf.approx_pendulum_callables(hidden=8, dtype=torch.float32)
    _roundtrip(f32, q32, 2, 1)


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
        code = trace.generate(f, q, 2, 2, term)
        # (a tensor the callable creates is a symbolic constant now: a dynamics that returns torch.zeros(1, 2) translates, and is
        # caught where every translation is checked -- against the callable on a batch)
        trace.verify_on_host(code, f, q, 2, 2, term)


def test_verification_catches_a_wrong_translation():
    f, q = jf.ref_pendulum_callables()
    code = trace.generate(f, q, 2, 1)
    code["step"] = code["step"].replace("T(0.05)", "T(0.06)")
    with pytest.raises(trace.TraceUnsupported):
        trace.verify_on_host(code, f, q, 2, 1)


def test_schedule_indexed_by_the_timestep():
    """`ref[t]`, `ref[t + 1, 0]`, `gain[t]` on constant tensors (trajectory tracking under step_dependent_dynamics): constant
    tables in the functor; the verification stays inside the horizon"""
    f, q = jf.tracking_callables(T=24)
    code = trace.generate(f, q, 2, 2, None, True)
    assert "tab" in code["cost"] and "tab" in code["step"]
    assert trace.verify_on_host(code, f, q, 2, 2, None, True, horizon=24)
    short_f, short_q = jf.tracking_callables(T=8)                  # tables of 8 / 9 entries: t = 11 would be out of range
    code = trace.generate(short_f, short_q, 2, 2, None, True)
    assert trace.verify_on_host(code, short_f, short_q, 2, 2, None, True, horizon=8)
    with pytest.raises(IndexError):
        trace.verify_on_host(code, short_f, short_q, 2, 2, None, True)
    # a table of traced values cannot be looked up (its entries are not constants)
    with pytest.raises(trace.TraceUnsupported):
        trace.generate(lambda s, a, t: s + torch.stack((s[0], a[0]))[t], lambda s, a, t: (s ** 2).sum(-1), 2, 2, None, True)


def test_symbolic_inputs_report_the_controllers_device_and_dtype():
    """`net.to(state.device, state.dtype)` inside a callable must not move the user's module while it is being traced"""
    seen = []

    def f(s, a):
        seen.append((s.device, s.dtype, s.is_cuda))
        return s + a

    q = lambda s, a: (s ** 2).sum(-1)
    trace.generate(f, q, 2, 2, device="cuda:0", dtype=torch.float32)
    assert seen[-1] == (torch.device("cuda:0"), torch.float32, True)
    trace.generate(f, q, 2, 2)
    assert seen[-1] == (torch.device("cpu"), torch.float64, False)
    # a float32 constant created from state.dtype enters the functor with its float32 value
    g = lambda s, a: s + a * torch.tensor(0.1, dtype=s.dtype)
    c32 = trace.generate(g, q, 2, 2, dtype=torch.float32)["step"]
    assert repr(float(torch.tensor(0.1, dtype=torch.float32))) in c32 and "T(0.1)" in trace.generate(g, q, 2, 2)["step"]


def test_constants_derived_inside_the_callable_are_watched_at_their_source():
    """`B.to(state.device)`, `W @ W.T`, `torch.diag(q)`: the functor holds the VALUES of the derived tensor; the version watch must
    sit on B / W / q -- the copies are never written again"""
    B = torch.tensor([[1.0, 0.0], [0.5, -1.0]], dtype=torch.float32)          # (another dtype: .to() makes a copy)
    W = torch.tensor([[1.0, 0.2], [0.0, 1.0]], dtype=torch.float64)
    qd = torch.tensor([1.0, 3.0], dtype=torch.float64)
    f = lambda s, a: s + a @ B.to(s.device, s.dtype).T
    q = lambda s, a: ((s @ (W @ W.T)) * s).sum(-1) + (s @ torch.diag(qd) * s).sum(-1)
    code = _roundtrip(f, q, 2, 2)
    roots = [t for t, _ in code["captured"]]
    assert any(t is B for t in roots) and any(t is W for t in roots) and any(t is qd for t in roots)
    assert all(t._version == v for t, v in code["captured"])
    B[0, 1] = 2.0
    assert any(t._version != v for t, v in code["captured"])


def test_two_outputs_from_the_same_input_component():
    """x'[0] = x'[1] = x[2] (found by differential fuzzing: the protective copy of x[2] was declared twice)"""
    f = lambda s, a: torch.stack((s[:, 2], s[:, 2], s[:, 0] + a[:, 0]), 1)
    _roundtrip(f, lambda s, a: (s ** 2).sum(-1), 3, 1)


def test_python_modulo_is_exact_at_multiples_of_the_modulus():
    """`floor(u) % 0.7` (found by differential fuzzing): k = floor(a / b) from the rounded quotient is off by one at exact
    multiples of b; torch's remainder is fmod + a sign fix-up.  A positive constant modulus goes through m_floormod."""
    f = lambda s, a: torch.stack((torch.floor(a[:, 0] * 3) % 0.7, (s[:, 1] * 4).round() % 0.1, s[:, 0] % (2 * math.pi)), 1)
    code = _roundtrip(f, lambda s, a: (s ** 2).sum(-1), 3, 1)
    assert code["step"].count("m_floormod") == 3
    import numpy as np
    X = np.array([[7.0, 0.25, 0.0], [0.7 * 3, 0.5, 0.0], [-1.4, -0.75, 0.0]])
    U = np.array([[7.0 / 3], [1.0], [-2.0]])
    got = trace.evaluate_on_host(code, X, U, 3, 1)[0]
    want = f(torch.tensor(X), torch.tensor(U)).numpy()
    assert np.array_equal(got, want), (got, want)


# ---------------------------------------------------------------------------------------------------------------------
# dense layers kept as layers (VERDICT r03 missing #2): F.linear on a real weight tensor becomes an MlpLayer member of the functor
# (csrc/mlp_wide.hpp: fma chains per lane, or matrix-core tiles of sixteen samples in the wide kernel); chains of layers joined
# by one elementwise activation keep their hidden activations in the distributed form
# ---------------------------------------------------------------------------------------------------------------------
def _check_against_torch(code, f, q, nx, nu, seed=0):
    g = torch.Generator().manual_seed(seed)
    X, U = torch.randn(9, nx, generator=g, dtype=torch.float64), torch.randn(9, nu, generator=g, dtype=torch.float64)
    xn, cc, _ = trace.evaluate_on_host(code, X.numpy(), U.numpy(), nx, nu)
    with torch.no_grad():
        assert np.allclose(xn, f(X, U).double().numpy(), rtol=1e-9, atol=1e-11)
        assert np.allclose(cc, q(X, U).double().reshape(-1).numpy(), rtol=1e-9, atol=1e-11)


def test_dense_layers_stay_layers_and_chain_through_their_activations():
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 2)).double()
    f = lambda s, a: s + net(torch.cat((s, a), 1))
    q = lambda s, a: (s ** 2).sum(1)
    code = jit.trace_and_verify(f, q, 2, 1)
    assert [(d["IN"], d["OUT"], d["kind"]) for d in code["dense"]] == [(3, 32, 0), (32, 32, 1), (32, 2, 1)]
    assert "mlp_first(ml0_0" in code["step"] and "mlp_mid(ml1_1" in code["step"] and "mlp_last(ml2_1" in code["step"]
    assert code["step"].count("m_tanh") == 2                 # the activations run over whole register arrays, not per element
    assert "MlpLayer<32, 32, 1, WX, true, T, ParamPtr> ml1_1;" in code["members"] and "ml1_1.load(p + 128, p + 1152);" in code["ctor"]
    assert code["n_params"] == 3 * 32 + 32 + 32 * 32 + 32 + 32 * 2 + 2
    _check_against_torch(code, f, q, 2, 1)


def test_dense_layer_chains_break_where_the_pattern_does():
    torch.manual_seed(4)
    l1, l2, l4 = torch.nn.Linear(4, 16).double(), torch.nn.Linear(16, 16).double(), torch.nn.Linear(16, 4).double()
    l3 = lambda h: l4(h)[:, :3]                # (16 x 4 = 64 multiply-adds: the smallest layer that stays a layer)

    def f(s, a):                               # a skip connection: h1 is used twice -> three layers on their own
        h1 = torch.relu(l1(torch.cat((s, a), 1)))
        h2 = torch.relu(l2(h1)) + h1
        return s + 0.1 * l3(h2)
    q = lambda s, a: (s ** 2).sum(1) + 0.1 * (a ** 2).sum(1)
    code = jit.trace_and_verify(f, q, 3, 1)
    assert code["step"].count("mlp_single(") == 3 and "mlp_mid" not in code["step"]
    _check_against_torch(code, f, q, 3, 1)

    def f2(s, a):                              # relu / sigmoid / a scale between the layers: one chain
        h = torch.sigmoid(l2(torch.relu(l1(torch.cat((s, a), 1)))))
        return s + l3(0.5 * h)
    code = jit.trace_and_verify(f2, q, 3, 1)
    assert "mlp_first(ml0_0" in code["step"] and "mlp_mid(ml1_1" in code["step"] and "mlp_last(ml2_1" in code["step"]
    assert "m_max(d0[i_], T(0.0))" in code["step"] and "* T(0.5)" in code["step"]
    _check_against_torch(code, f2, q, 3, 1)

    small = torch.nn.Linear(3, 4).double()     # 12 multiply-adds: scalar terms, no layer
    code = jit.trace_and_verify(lambda s, a: s + small(s)[:, :3] * a, q, 3, 1)
    assert not code["dense"] and "mlp_" not in code["step"]

    # the same layer applied to two different inputs: two calls on one member
    def f3(s, a):
        return s + 0.1 * l3(torch.tanh(l2(torch.tanh(l1(torch.cat((s, a), 1)))))) - 0.1 * l3(torch.tanh(l2(torch.tanh(l1(torch.cat((-s, a), 1))))))
    code = jit.trace_and_verify(f3, q, 3, 1)
    assert code["step"].count("mlp_first(ml0_0") == 1 and code["step"].count("mlp_first(ml3_0") == 1     # (two applications = two layer records)
    _check_against_torch(code, f3, q, 3, 1)


def test_dense_layers_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("MPPI_TRACE_DENSE", "0")
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2)).double()
    code = jit.trace_and_verify(lambda s, a: s + net(torch.cat((s, a), 1)), lambda s, a: (s ** 2).sum(1), 2, 1)
    assert not code["dense"] and "mlp_" not in code["step"] and code["members"] == ""


def test_a_trace_that_is_the_matrix_core_mlp_is_recognised():
    """VERDICT r04 item 4 (first half): torch callables whose trace is x + s (W2 tanh(W1 [x; u] + b1) + b2) with cost sum x^2 --
    BASELINE configs[3] written as the reference's plugin API -- are matched structurally (trace.match_mlp_residual) so that the
    controller can hand them to the hand-written matrix-core kernel instead of compiling a functor"""
    import torch
    from pytorch_mppi_amd import jit, trace
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(20, 64), torch.nn.Tanh(), torch.nn.Linear(64, 16)).double()
    dyn = lambda x, u: x + 0.1 * net(torch.cat((x, u), -1))
    sq = lambda x, u: (x ** 2).sum(-1)
    code = trace.generate(dyn, sq, 16, 4)
    sp = code["mlp_residual"]
    assert sp == dict(H=64, w1=0, b1=1280, w2=1344, b2=2368, scale=0.1)
    # the parameter blob the kernel wants, out of the trace's parameter vector
    m = jit.TracedMLPResidual("t", 16, 4, dyn, sq, sp, trace.gather_params(code["param_tensors"], code["n_params"]))
    W1, b1, W2, b2, s = m._param_list()[:5]
    assert torch.equal(W1.reshape(64, 20), net[0].weight.detach()) and torch.equal(b1, net[0].bias.detach())
    assert torch.equal(W2.reshape(16, 64), net[2].weight.detach()) and torch.equal(b2, net[2].bias.detach()) and float(s) == 0.1
    assert m.model_id == 4 and m.hidden == 64 and m.flags() == 0
    # a width between the kernels' widths: zero-padded units (exactly nothing added), the kernel sees the next width
    net100 = torch.nn.Sequential(torch.nn.Linear(20, 100), torch.nn.Tanh(), torch.nn.Linear(100, 16)).double()
    dyn100 = lambda x, u: x + 0.1 * net100(torch.cat((x, u), -1))
    c100 = trace.generate(dyn100, sq, 16, 4)
    m100 = jit.TracedMLPResidual("t100", 16, 4, dyn100, sq, c100["mlp_residual"], trace.gather_params(c100["param_tensors"], c100["n_params"]))
    P = m100._param_list()
    assert (m100.hidden, m100.hidden_units) == (128, 100) and [t_.numel() for t_ in P] == [128 * 20, 128, 16 * 128, 16, 1, 16, 4]
    assert torch.equal(P[0][:100], net100[0].weight.detach()) and not P[0][100:].any() and not P[1][100:].any() and not P[2][:, 100:].any()
    assert torch.equal(P[2][:, :100], net100[2].weight.detach())
    x_, u_ = torch.randn(5, 16, dtype=torch.float64), torch.randn(5, 4, dtype=torch.float64)
    padded = x_ + 0.1 * (torch.tanh(torch.cat((x_, u_), -1) @ P[0].T + P[1]) @ P[2].T + P[3])
    assert torch.allclose(padded, dyn100(x_, u_), rtol=0, atol=1e-15)
    # what is NOT that shape keeps its functor
    # a diagonal quadratic cost with control effort is the kernel's too (round 5): weights end up in the blob behind res_scale
    qw = torch.linspace(0.5, 2.0, 16, dtype=torch.float64)
    code_q = trace.generate(dyn, lambda x, u: (qw * x ** 2).sum(-1) + 0.1 * (u ** 2).sum(-1), 16, 4)
    spq = code_q["mlp_residual"]
    assert spq is not None and spq["qu"] == [0.1] * 4 and max(abs(a - b) for a, b in zip(spq["qx"], qw.tolist())) < 1e-15
    mq = jit.TracedMLPResidual("tq", 16, 4, dyn, sq, spq, trace.gather_params(code_q["param_tensors"], code_q["n_params"]))
    assert torch.equal(mq._param_list()[5], qw) and mq._param_list()[6].tolist() == [0.1] * 4
    assert [t_.numel() for t_ in m._param_list()] == [1280, 64, 1024, 16, 1, 16, 4] and m._param_list()[5].tolist() == [1.0] * 16
    for d_, c_ in ((dyn, lambda x, u: (x ** 2).sum(-1) + x[..., 0] * x[..., 1]),                     # a cross term: no diagonal form
                   (dyn, lambda x, u: x.abs().sum(-1)),                                              # another cost altogether
                   (lambda x, u: x + 0.1 * net(torch.cat((u, x), -1)), sq),                          # inputs in another order
                   (lambda x, u: 0.9 * x + 0.1 * net(torch.cat((x, u), -1)), sq),                    # no plain residual
                   (lambda x, u: x + 0.1 * torch.nn.functional.relu(net[0](torch.cat((x, u), -1))) @ net[2].weight.T, sq)):   # another activation
        assert trace.generate(d_, c_, 16, 4)["mlp_residual"] is None
    two = torch.nn.Sequential(torch.nn.Linear(20, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 16)).double()
    assert trace.generate(lambda x, u: x + two(torch.cat((x, u), -1)), sq, 16, 4)["mlp_residual"] is None
    assert trace.generate(lambda x, u, t: x + 0.1 * net(torch.cat((x, u), -1)), lambda x, u, t: (x ** 2).sum(-1), 16, 4, step_dependent=True)["mlp_residual"] is None
