"""Evidence guard: bench.py's line quotes counter passes committed under profiles/ (pmc_*.json) -- the
on-chip kernel's VALU counters and HBM traffic, the streaming K1's traffic, the C4 kernel's matrix-pipe share.  Each entry is
stamped with the sha256 of the kernel's translation unit (its source, every header it includes, its flags) as it was when the
counters were collected.  An entry whose kernel has changed since is not evidence for the kernel that is timed today: bench.py flags it
(`lookup_stale`), and THIS test fails -- re-collect (tools/rounds/*_final.sh, tools/pmc_lookup_update.py) before the round's head."""
import json
import os

from pytorch_mppi_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _entries():
    """(file, where, kernel, entry) of the lookups bench.py's default line reads"""
    out = []
    v = json.load(open(os.path.join(ROOT, "profiles", "pmc_onchip_valu.json")))
    out.append(("pmc_onchip_valu.json", "c3/philox-onchip", "rollout_onchip_kernel", v["c3/philox-onchip"]["rollout_onchip_kernel"]))
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, kernel in (("c3/philox-onchip", "rollout_onchip_kernel"), ("c3/philox-stream", "rollout_cost_kernel")):
        out.append(("pmc_traffic.json", key, kernel, t[key][kernel]))
    out.append(("pmc_c4_mfma.json", "", "rollout_mlp_split_kernel", json.load(open(os.path.join(ROOT, "profiles", "pmc_c4_mfma.json")))))
    return out


def test_every_kernel_of_the_guard_has_a_unit_and_a_hash():
    for k, unit in _build.KERNEL_UNITS.items():
        assert os.path.exists(os.path.join(_build.CSRC, unit)), (k, unit)
        h = _build.kernel_sources_hash(k)
        assert len(h) == 64 and h == _build.kernel_sources_hash(k)
    # the hash follows the headers a unit includes, not only its own file
    closure = [os.path.basename(p) for p in _build._include_closure("rollout_integrator.hip")]
    assert {"rollout.hpp", "rollout_onchip.hpp", "common.hpp", "mppi_amd.h"} <= set(closure), closure


def test_committed_lookups_describe_the_kernels_of_this_tree():
    stale = []
    for fname, where, kernel, e in _entries():
        want = _build.kernel_sources_hash(kernel)
        if e.get("sources_sha256") != want:
            stale.append(f"profiles/{fname} {where} {kernel}: collected on {e.get('collected', '?')} from sources {str(e.get('sources_sha256'))[:12]}..., "
                         f"the tree's are {want[:12]}...")
    assert not stale, "counter passes quoted by bench.py are stale -- re-collect them (tools/rounds/r06_final.sh + tools/pmc_lookup_update.py):\n" + "\n".join(stale)


def test_bench_flags_a_stale_lookup():
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    bench.LOOKUPS_USED.clear()
    good = {"sources_sha256": _build.kernel_sources_hash("rollout_onchip_kernel")}
    assert bench._lookup_stale(good, "rollout_onchip_kernel", "x") is False
    assert bench._lookup_stale({"sources_sha256": "0" * 64}, "rollout_onchip_kernel", "y") is True
    assert bench._lookup_stale({}, "rollout_onchip_kernel", "z") is True            # no hash on record: not evidence
    assert bench.LOOKUPS_USED == {"x": False, "y": True, "z": True}
