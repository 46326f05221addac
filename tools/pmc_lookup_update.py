"""profiles/pmc_onchip_valu.json (the lookup bench.py quotes as roofline.frac / roofline.traffic of the on-chip kernel) and
profiles/pmc_c4_mfma.json from the per-counter tables a round's PMC passes left behind:
    python tools/pmc_lookup_update.py gpurun_out/r05_final      # reads <prefix>_pmc_c3_{valu,wait,fetch,write}.txt, _pmc_c4_mfma.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path, kernel):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if m and kernel in m.group(6):
            out[m.group(1)] = float(m.group(3))
    return out


def main(prefix):
    tag = os.path.basename(prefix)
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
    d["_comment"] = (f"SQ counters of the on-chip K1 (rollout_onchip_kernel<Integrator<16,12>>, K = 65536 = 1024 waves, one per SIMD) from separate rocprofv3 --pmc "
                     f"passes of `bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline` on MI355X: profiles/{tag}_pmc_c3_valu.txt, _wait.txt, _fetch.txt, "
                     f"_write.txt (instruction classes: profiles/r04_spill_pmc_c3_classes.txt, the same kernel code).  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count "
                     f"quad-cycles summed over the waves (MI355X_MICROARCH.md); means over the dispatches of the pass.")
    json.dump(d, open(p, "w"), indent=1)
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    t.setdefault("c3/philox-onchip", {})[k] = {"FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": c["WRITE_SIZE"],
                                                "traffic_bytes": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
                                                "source": f"profiles/{tag}_pmc_c3_fetch.txt / _write.txt"}
    json.dump(t, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    m = table(f"{prefix}_pmc_c4_mfma.txt", "rollout_mlp_split_kernel")
    if m:
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        q = os.path.join(ROOT, "profiles", "pmc_c4_mfma.json")
        j = json.load(open(q))
        j["mfma_busy"] = round(busy, 4)
        j["_comment"] = (f"C4 (rollout_mlp_split_kernel<16,0,true>, K=65536, T=64, H=256): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) "
                         f"= {m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} / ({m['GRBM_GUI_ACTIVE']:.0f} / 8 x 1024) from profiles/{tag}_pmc_c4_mfma.txt; valu_issue_frac = estimated VALU issue "
                         f"cycles of the hot loop (profiles/r04_c4_isa_issue_budget.txt) x iterations / measured kernel cycles.")
        j["source"] = f"profiles/pmc_c4_mfma.json <- profiles/{tag}_pmc_c4_mfma.txt and profiles/r04_c4_isa_issue_budget.txt; lookups, not measured in this run"
        json.dump(j, open(q, "w"), indent=1)
    print("VALU-active share", e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"], "traffic MB", (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6, "c4 busy", m and busy)


if __name__ == "__main__":
    main(sys.argv[1])
