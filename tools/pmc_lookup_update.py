"""profiles/pmc_onchip_valu.json (the lookup bench.py quotes as roofline.frac / roofline.traffic of the on-chip kernel) and
profiles/pmc_c4_mfma.json from the per-counter tables a round's PMC passes left behind:
    python tools/pmc_lookup_update.py gpurun_out/r06_final      # reads <prefix>_pmc_c3_{valu,wait,fetch,write}.txt, _pmc_c4_mfma.txt,
                                                                # _pmc_stream_{fetch,write}.txt, _pmc_torch_{fetch,write}.txt
Every refreshed entry is stamped with `sources_sha256` = pytorch_mppi_amd/_build.kernel_sources_hash(kernel) AS OF THE COLLECTION
(<prefix>_kernel_source_hashes.json, written on the GPU box by the script that ran the passes; this tree's hash if that file is
missing): bench.py prints `lookup_stale` and tests/test_lookup_evidence.py fails when the tree has moved on."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(path, kernel):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        # (the on-chip K1 is rollout_onchip_kernel or, round 6, rollout_onchip_pair_kernel: whichever the pass ran)
        if m and (kernel in m.group(6) or (kernel == "rollout_onchip_kernel" and "rollout_onchip_pair_kernel" in m.group(6))):
            out[m.group(1)] = float(m.group(3))
    return out


def hashes_of(prefix):
    from pytorch_mppi_amd import _build
    now = {k: _build.kernel_sources_hash(k) for k in _build.KERNEL_UNITS}
    p = f"{prefix}_kernel_source_hashes.json"
    if os.path.exists(p):
        then = json.load(open(p))
        moved = [k for k in now if then.get(k) != now[k]]
        if moved:
            print("NOTE: kernel sources changed since these passes were collected:", ", ".join(moved), "-- the entries are stamped with the hashes "
                  "of the collection and will read as stale")
        return then
    return now


def main(prefix):
    tag = os.path.basename(prefix)
    H = hashes_of(prefix)
    k = "rollout_onchip_kernel"
    c = {}
    for part in ("valu", "wait", "fetch", "write"):
        c.update(table(f"{prefix}_pmc_c3_{part}.txt", k))
    p = os.path.join(ROOT, "profiles", "pmc_onchip_valu.json")
    d = json.load(open(p))
    e = d["c3/philox-onchip"][k]
    for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
        if name in c:
            e[name] = c[name]
    e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
    # which on-chip K1 the passes ran: the two-wave kernel (round 6) has 2048 waves at C3, two per SIMD
    pair = "rollout_onchip_pair_kernel" in open(f"{prefix}_pmc_c3_valu.txt").read()
    e["kernel"] = "rollout_onchip_pair_kernel" if pair else "rollout_onchip_kernel"
    e["waves"], e["waves_per_simd"] = (2048, 2) if pair else (1024, 1)
    e["sources_sha256"], e["collected"] = H[k], tag
    d["_comment"] = (f"SQ counters of the on-chip K1 (the kernel the default bench ran: rollout_onchip_pair_kernel<Integrator<16,12>> since round 6 -- K = 65536 = 2048 waves, two per SIMD -- rollout_onchip_kernel before) from separate rocprofv3 --pmc "
                     f"passes of `bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline` on MI355X: profiles/{tag}_pmc_c3_valu.txt, _wait.txt, _fetch.txt, "
                     f"_write.txt (instruction classes `by_class`: profiles/r04_spill_pmc_c3_classes.txt, an earlier build of the kernel).  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count "
                     f"quad-cycles summed over the waves (MI355X_MICROARCH.md); means over the dispatches of the pass.  sources_sha256: the kernel's "
                     f"translation unit when the passes ran (pytorch_mppi_amd/_build.kernel_sources_hash).")
    json.dump(d, open(p, "w"), indent=1)
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    t = json.load(open(tp))

    def traffic(key, kernel, fetch_file, write_file, alg=None):
        f, w = table(fetch_file, kernel), table(write_file, kernel)
        if "FETCH_SIZE" not in f or "WRITE_SIZE" not in w:
            print("no counters for", key, kernel)
            return
        t.setdefault(key, {})[kernel] = {"FETCH_SIZE": f["FETCH_SIZE"], "WRITE_SIZE": w["WRITE_SIZE"],
                                         "traffic_bytes": int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024),
                                         **({"algorithmic_bytes": alg} if alg else {}),
                                         "source": f"profiles/{os.path.basename(fetch_file)} / {os.path.basename(write_file)}",
                                         "sources_sha256": H[kernel], "collected": tag}
    B1 = 4 * 65536 * 64 * 12 + 4 * 65536
    traffic("c3/philox-onchip", k, f"{prefix}_pmc_c3_fetch.txt", f"{prefix}_pmc_c3_write.txt")
    for key, name in (("c3/philox-stream", "stream"), ("c3/torch", "torch")):
        if os.path.exists(f"{prefix}_pmc_{name}_fetch.txt"):
            traffic(key, "rollout_cost_kernel", f"{prefix}_pmc_{name}_fetch.txt", f"{prefix}_pmc_{name}_write.txt", B1)
            traffic(key, "weights_partial_rows_kernel", f"{prefix}_pmc_{name}_fetch.txt", f"{prefix}_pmc_{name}_write.txt")
            traffic(key, "noise_fill_philox_kernel", f"{prefix}_pmc_{name}_fetch.txt", f"{prefix}_pmc_{name}_write.txt")
    json.dump(t, open(tp, "w"), indent=1)
    m = table(f"{prefix}_pmc_c4_mfma.txt", "rollout_mlp_split_kernel") if os.path.exists(f"{prefix}_pmc_c4_mfma.txt") else {}
    busy = None
    if m:
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        q = os.path.join(ROOT, "profiles", "pmc_c4_mfma.json")
        j = json.load(open(q))
        j["mfma_busy"] = round(busy, 4)
        j["sources_sha256"], j["collected"] = H["rollout_mlp_split_kernel"], tag
        j["_comment"] = (f"C4 (rollout_mlp_split_kernel<16,0,true>, K=65536, T=64, H=256): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) "
                         f"= {m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} / ({m['GRBM_GUI_ACTIVE']:.0f} / 8 x 1024) from profiles/{tag}_pmc_c4_mfma.txt; valu_issue_frac = estimated VALU issue "
                         f"cycles of the hot loop (profiles/r04_c4_isa_issue_budget.txt) x iterations / measured kernel cycles.")
        j["source"] = f"profiles/pmc_c4_mfma.json <- profiles/{tag}_pmc_c4_mfma.txt and profiles/r04_c4_isa_issue_budget.txt; lookups, not measured in this run"
        json.dump(j, open(q, "w"), indent=1)
    print("VALU-active share", e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"], "traffic MB", (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6, "c4 busy", busy)


if __name__ == "__main__":
    main(sys.argv[1])
