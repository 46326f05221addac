"""CPU: the oracle restatement (oracle/mppi_oracle.py) against the committed fixtures that the
LIVE reference produced (oracle/gen_golden.py).  This is what pins the oracle on machines that
do not have /root/reference."""
import numpy as np
import pytest
import torch

import golden_util as gu

KEYS = ["action", "U", "cost_total", "omega", "noise", "perturbed_action"]


@pytest.mark.parametrize("name", gu.golden_names())
def test_oracle_matches_reference_fixture(name):
    cfg, d = gu.load(name)
    outs = gu.oracle_run(cfg, d)
    # same torch ops in the same order => identical up to the last bit on the same torch build;
    # tolerances only absorb BLAS/vectorisation differences between CPU models.
    rtol = 1e-12 if cfg["dtype"] == "f64" else 2e-6
    if cfg["kmppi"]:
        rtol = 1e-9 if cfg["dtype"] == "f64" else 1e-4   # constant-W form vs vmap(solve), SURVEY 3.3
    for s, r in enumerate(outs):
        for k in KEYS + (["theta", "noise_theta"] if cfg["kmppi"] else []) + (["action_sequence"] if cfg.get("smppi") else []):
            ref = np.array(d[f"{k}{s}"])
            got = r[k].numpy()
            scale = max(1.0, float(np.abs(ref).max()))
            np.testing.assert_allclose(got, ref, rtol=rtol, atol=rtol * scale, err_msg=f"{name} step {s} {k}")
        if cfg["sampler_rows"]:
            assert tuple(d[f"slice{s}"]) == tuple(r["sampler_slice"])      # bit-exact index bookkeeping


def test_omega_sums_to_one():
    """reference test_mppi.py:269-274"""
    cfg, d = gu.load("linear_diag_f64")
    r = gu.oracle_run(cfg, d)[0]
    assert abs(float(r["omega"].sum()) - 1.0) < 1e-12


def test_rbf_kernel_known_answers():
    """the reference's only numeric KAT, test_mppi.py:560-570: diag = 1, off-diag = e^-0.5"""
    from oracle.mppi_oracle import rbf_kernel
    tt = torch.tensor([[0.0], [1.0]], dtype=torch.double)
    k = rbf_kernel(tt, tt, sigma=1.0)
    assert torch.allclose(k.diag(), torch.ones(2, dtype=torch.double), atol=1e-6)
    assert torch.allclose(k[0, 1], torch.tensor(np.exp(-0.5), dtype=torch.double), atol=1e-6)


@pytest.mark.parametrize("name", gu.golden_names(batched=True))
def test_oracle_batched_matches_reference_fixture(name):
    """MPPI_Batched (mppi.py:691-873) == N independent oracle commands sharing one z."""
    cfg, d = gu.load(name)
    outs = gu.oracle_run_batched(cfg, d)
    rtol = 1e-12 if cfg["dtype"] == "f64" else 2e-6
    for s, r in enumerate(outs):
        for k in ("action", "U"):
            ref = np.array(d[f"{k}{s}"])
            np.testing.assert_allclose(r[k].numpy(), ref, rtol=rtol, atol=rtol * max(1.0, np.abs(ref).max()),
                                       err_msg=f"{name} step {s} {k}")
