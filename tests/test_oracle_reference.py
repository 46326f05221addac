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
