"""Register / scratch budgets of the hot kernels, read from the code objects of the built library (no GPU needed).

A kernel that starts spilling is a performance regression no parity test sees: in round 3 the KMPPI-fused K1 went from
20 B to 528 B of scratch per lane (131 spilled VGPRs: 70.7 -> 94 us, the whole KMPPI command +19 %) when the Philox rounds
moved to v_bitop3_b32, and nobody noticed.  The budgets below are what the kernels are DESIGNED to
hold: the on-chip command and the C3 streaming K1 / K3 run one wave per SIMD out of registers alone, the split
MLP kernel keeps its layer-1 weights pinned in AGPRs, the KMPPI-fused K1 keeps 256 control points there."""
import os

import pytest

from pytorch_mppi_amd import _build, _resources as R


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(_build.LIB):
        pytest.skip("library not built")
    return R.kernels(_build.LIB)


def _one(table, *needles):
    hit = R.find(table, *needles)
    assert len(hit) == 1, (needles, sorted(hit))
    return next(iter(hit.values()))


# (what to find in the demangled name, max scratch bytes per lane, max spilled VGPRs)
BUDGETS = [
    # the headline command: on-chip K1 (C3 and the other fp32 integrator / pendulum / linear shapes)
    (("rollout_onchip_kernel<mppi::IntegratorModel<float, 16, 12>, true, false>",), 0, 0),
    (("rollout_onchip_kernel<mppi::IntegratorModel<float, 16, 12>, true, true>",), 0, 0),      # rng="philox7"
    (("rollout_onchip_kernel<mppi::PendulumModel<float>, true, false>",), 0, 0),
    (("rollout_onchip_kernel<mppi::LinearGoalModel<float, 12, 4>, true, false>",), 0, 0),
    (("finalize_blocks_kernel<float>",), 0, 0),
    # streaming K1 at C3 (rows from memory; diagonal Sigma; the plain and the select-only variants run per wave)
    (("rollout_cost_kernel<mppi::IntegratorModel<float, 16, 12>, float, 0, true, 0, false, 1>",), 0, 0),
    # K3 / K4 of the streaming command
    (("weights_partial_diag_kernel<float, 0,",), 0, 0),
    (("finalize_kernel<float>",), 0, 0),
    # the generator launch
    (("noise_fill_philox_kernel<float>",), 0, 0),
    # KMPPI with the interpolation inside K1: rows generated in the prologue (1) / read from memory (0)
    (("rollout_kmppi_kernel<mppi::IntegratorModel<float, 16, 12>, 1>",), 20, 0),
    (("rollout_kmppi_kernel<mppi::IntegratorModel<float, 16, 12>, 0>",), 20, 0),
    (("rollout_kmppi_kernel<mppi::LinearGoalModel<float, 12, 4>, 1>",), 20, 0),
    # the MLP on the matrix cores (C4 / C5): split-operand kernel, hidden 256
    (("rollout_mlp_split_kernel<16, 0, true, 16, 4>",), 0, 0),      # C4
    (("rollout_mlp_split_kernel<16, 0, true, 12, 6>",), 0, 0),      # the further (nx, nu) of the split-operand kernel: no scratch either
    (("rollout_mlp_split_kernel<16, 0, true, 8, 2>",), 0, 0),
    (("rollout_mlp_split_kernel<16, 0, true, 16, 8>",), 0, 0),
]


@pytest.mark.parametrize("needles,scratch,spill", BUDGETS, ids=[b[0][0][:70] for b in BUDGETS])
def test_hot_kernels_stay_within_their_register_budgets(table, needles, scratch, spill):
    hit = R.find(table, *needles)
    assert hit, f"no kernel matching {needles} in the library: renamed? update the budget table"
    for name, r in hit.items():
        assert r.get("scratch", 0) <= scratch, f"{name}: {r['scratch']} B of scratch per lane (budget {scratch}): {r}"
        assert r.get("vgpr_spill", 0) <= spill, f"{name}: {r['vgpr_spill']} spilled VGPRs (budget {spill}): {r}"


def test_no_fp32_kernel_of_the_path_spills_vector_registers(table):
    """every fp32 kernel of the engine: no VGPR spills at all (fp64 instantiations of the widest models are allowed a few:
    they are the parity yardstick, not a performance path)"""
    # (weights_partial_rows_kernel<4>: held at 128 registers -- four waves per SIMD for the launch that also generates the next
    #  draw, csrc/noise_torch.hip -- at the price of ONE value spilled in front of a barrier, outside the streaming loop)
    allowed = {"weights_partial_rows_kernel<4>": 2}
    bad = {k: v for k, v in table.items() if v.get("vgpr_spill", 0) > max([n for a_, n in allowed.items() if a_ in k] + [0]) and "double" not in k}
    assert not bad, bad


def test_resource_table_sees_the_whole_library(table):
    assert len(table) > 300
    assert R.find(table, "rollout_onchip_kernel") and R.find(table, "rollout_mlp_split_kernel") and R.find(table, "combine_kernel")


def test_run_time_compiled_models_carry_no_scratch_to_speak_of():
    """the functors __graft_entry__.build() compiles from the fixtures' callables (pytorch_mppi_amd/_jit): what the scalar vocabulary
    of traced code (csrc/common.hpp m_sin, m_tanh, ...) drags into a user's kernels shows here -- a library sinf in m_sin's rare path
    once brought 400 B of scratch per lane with it.  fp32 kernels: no VGPR spills, at most 64 B of scratch."""
    import glob
    import os
    objs = sorted(glob.glob(os.path.join(os.path.dirname(R.__file__), "_jit", "user_*.so")), key=os.path.getmtime)[-24:]
    if not objs:
        pytest.skip("no run-time compiled models here (run __graft_entry__.build())")
    bad = {}
    seen = 0
    for f in objs:
        for k, v in R.kernels(f).items():
            if "double" in k:
                continue
            seen += 1
            if v.get("scratch", 0) > 64 or v.get("vgpr_spill", 0) > 0:
                bad[os.path.basename(f) + " :: " + k[:90]] = v
    assert seen > 20 and not bad, bad
