"""Seed distributions of the parity margins that exceed the north-star's literal 1e-5.

    python tools/margin_distributions.py [n_seeds=32] [out_prefix=gpurun_out/r06_margin_distributions]

The -m gpu suite records, per comparison, err(engine fp32 vs fp64 oracle) / scale and the reference's OWN fp32 error against the
same fp64 run (the floor; tests/margins.py).  SURVEY 7.3 passes an entry above 1e-5 at up to 2 x its floor.  Thirty entries of the
round-5 ledger were above 1e-5, each measured on ONE seed.  This tool re-runs the scenarios those entries come from on `n_seeds`
different draws (MPPI_MARGIN_SEED shifts the initial state, the nominal sequence and the generator key; the model weights stay)
-- the MLP ones under both matrix-core kernels (rollout_mlp_split, the product's; MPPI_MLP_EXACT=1, the exact-fp32 checker) -- and
prints, per (scenario, quantity, kernel): how many seeds put the entry above 1e-5 at all, and median / p95 / max of
err / OWN floor over all seeds.  The test functions themselves are called (their assertions included: a seed that fails SURVEY
7.3's 2 x floor is listed as FAILED)."""
import json
import os
import re
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import margins  # noqa: E402
import test_gpu_fullsize_parity as F  # noqa: E402
import test_gpu_onchip as O  # noqa: E402
import test_gpu_random_configs as R  # noqa: E402


def scenarios():
    pend = [c for c in O.CASES if c[0] == "pendulum"][0]
    both = ("split", "exact")
    return [
        ("c2 pendulum healthy", lambda s: F.test_c2_pendulum_8192x32_philox_in_k1("healthy"), ("-",)),
        ("c2 pendulum peaked", lambda s: F.test_c2_pendulum_8192x32_philox_in_k1("peaked"), ("-",)),
        ("onchip pendulum K20000 T48", lambda s: O.test_onchip_command_matches_streaming_command_and_fp64_oracle(pend), ("-",)),
        ("c4 mlp healthy", lambda s: F.test_c4_mlp_65536x64_philox_generator_mfma("healthy"), both),
        ("c4 mlp peaked", lambda s: F.test_c4_mlp_65536x64_philox_generator_mfma("peaked"), both),
        ("smppi mlp H256", lambda s: F.test_smppi_with_mlp_dynamics_runs_on_the_matrix_cores(16, 4, 256, None), both),
        ("smppi mlp H64", lambda s: F.test_smppi_with_mlp_dynamics_runs_on_the_matrix_cores(16, 4, 64, None), both),
        ("c5 mlp 8 shards", lambda s: F.test_c5_eight_shards_of_the_mlp_equal_oracle_on_the_global_draw(), both),
        # the callback path's random configurations are a seed sweep by construction (its worst r05 entry was its seed 17)
        ("generic random config", lambda s: R._generic_vs_oracle(5000 + s, R._case(5000 + s)), ("-",)),
    ]


def main():
    F.ORACLE_DEVICE = os.environ.get("MARGIN_ORACLE_DEVICE", "cuda")     # (the full-size oracle runs: 25 s per seed of C4 on the host cores)
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    prefix = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r06_margin_distributions")
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    failures, timing = [], {}
    for name, fn, kernels in scenarios():
        if only and only not in name:
            continue
        for kern in kernels:
            t0 = time.time()
            if kern == "exact":
                os.environ["MPPI_MLP_EXACT"] = "1"
            else:
                os.environ.pop("MPPI_MLP_EXACT", None)
            for s in range(n_seeds):
                os.environ["MPPI_MARGIN_SEED"] = str(s)
                margins.TAG = [name, kern, s]
                try:
                    fn(s)
                except AssertionError as e:
                    failures.append((name, kern, s, str(e)[:300]))
                except Exception as e:          # a scenario that cannot run on this seed (e.g. N_eff outside its regime's window)
                    failures.append((name, kern, s, f"{type(e).__name__}: {str(e)[:300]}"))
                torch.cuda.synchronize()
            timing[(name, kern)] = time.time() - t0
            # (a long run that is cut off must not take everything with it: the tagged ledger so far, after every scenario)
            json.dump({"n_seeds": n_seeds, "done": [list(k) for k in timing], "ledger": [e for e in margins._LEDGER if e.get("tag")],
                       "failures": failures}, open(prefix + ".partial.json", "w"))
            print(f"[{time.strftime('%H:%M:%S')}] {name} / {kern}: {n_seeds} seeds in {timing[(name, kern)]:.0f} s", flush=True)
    os.environ.pop("MPPI_MLP_EXACT", None)
    os.environ.pop("MPPI_MARGIN_SEED", None)

    # ---- aggregate: (scenario, quantity, kernel) -> per-seed worst err/scale, err/own floor ----
    groups = {}
    for e in margins._LEDGER:
        tag = e.get("tag")
        if not tag or e.get("rtol") is None or e.get("floor_over_scale") is None:
            continue
        q = re.sub(r"seed \d+ ", "", e["quantity"])
        tname = e["test"]
        if tag[0].startswith("onchip") or tag[0].startswith("c2") or tag[0].startswith("c4"):
            q = tname.split("/")[-1] + " " + q if tag[0].startswith("onchip") else q
        g = groups.setdefault((tag[0], q, tag[1]), {})
        cur = g.get(tag[2])
        ratio = e["err_over_scale"] / e["floor_over_scale"] if e["floor_over_scale"] > 0 else float("nan")
        if cur is None or e["err_over_scale"] > cur[0]:
            g[tag[2]] = (e["err_over_scale"], e["floor_over_scale"], ratio)
    rows = []
    for (scen, q, kern), per_seed in sorted(groups.items()):
        errs = np.array([v[0] for v in per_seed.values()])
        ratios = np.array([v[2] for v in per_seed.values()])
        above = errs > 1e-5
        rows.append(dict(scenario=scen, quantity=q, kernel=kern, seeds=len(errs), seeds_above_1e5=int(above.sum()),
                         max_err_over_scale=float(errs.max()), median_ratio=float(np.nanmedian(ratios)),
                         p95_ratio=float(np.nanpercentile(ratios, 95)), max_ratio=float(np.nanmax(ratios)),
                         max_ratio_among_above=float(np.nanmax(ratios[above])) if above.any() else None,
                         worst_seed=int(list(per_seed)[int(np.nanargmax(np.where(above, ratios, -1.0)))]) if above.any() else None))
    lines = [f"# tools/margin_distributions.py: {n_seeds} seeds per scenario; err = max |engine fp32 - oracle fp64| / scale, floor = the oracle's own fp32 "
             "run against its fp64 run on the same draw (the full-size oracle runs with its tensors on the GPU: ATen kernels, MARGIN_ORACLE_DEVICE)",
             "# ratio = err / OWN floor (no sibling rule); SURVEY 7.3 passes err <= max(1e-5, 2 x floor); budget of tests/test_zz_margin_budget.py: 1.5",
             "# only (scenario, quantity, kernel) rows with at least one seed above 1e-5 are listed; `>1e-5` = seeds above / seeds run; "
             "max* = max ratio among the seeds above 1e-5 (what the budget test looks at)",
             f"# {'scenario':<28} {'quantity':<28} {'kernel':<6} {'>1e-5':>7} {'max err/scale':>13} {'median':>7} {'p95':>6} {'max':>6} {'max*':>6} {'seed':>5}"]
    for r in rows:
        if r["seeds_above_1e5"] == 0:
            continue
        lines.append(f"  {r['scenario']:<28} {r['quantity'][:28]:<28} {r['kernel']:<6} {r['seeds_above_1e5']:3d}/{r['seeds']:<3d} {r['max_err_over_scale']:13.3e} "
                     f"{r['median_ratio']:7.2f} {r['p95_ratio']:6.2f} {r['max_ratio']:6.2f} {r['max_ratio_among_above']:6.2f} {r['worst_seed']:5d}")
    quiet = [r for r in rows if r["seeds_above_1e5"] == 0]
    lines.append(f"# {len(quiet)} further (scenario, quantity, kernel) rows never exceeded 1e-5 on any seed (worst of them: "
                 f"{max((r['max_err_over_scale'] for r in quiet), default=0.0):.2e})")
    # split against exact, same scenario and quantity
    lines.append("# split-operand kernel against the exact-fp32 kernel (median of err / own floor over the seeds):")
    by = {(r["scenario"], r["quantity"], r["kernel"]): r for r in rows}
    for (scen, q, kern), r in sorted(by.items()):
        if kern != "split" or (scen, q, "exact") not in by:
            continue
        x = by[(scen, q, "exact")]
        if r["seeds_above_1e5"] or x["seeds_above_1e5"]:
            lines.append(f"#   {scen:<28} {q[:28]:<28} split median {r['median_ratio']:5.2f} max {r['max_ratio']:5.2f}   exact median {x['median_ratio']:5.2f} max {x['max_ratio']:5.2f}")
    lines.append(f"# FAILED (assertion of the test function itself, i.e. beyond 2 x floor, or scenario not runnable on that seed): {len(failures)}")
    for f in failures[:40]:
        lines.append(f"#   {f[0]} / {f[1]} / seed {f[2]}: {f[3]}".replace("\n", " ")[:400])
    lines.append("# wall time per scenario (s): " + ", ".join(f"{k[0]}/{k[1]} {v:.0f}" for k, v in timing.items()))
    txt = "\n".join(lines) + "\n"
    print(txt)
    open(prefix + ".txt", "w").write(txt)
    json.dump({"n_seeds": n_seeds, "rows": rows, "failures": failures}, open(prefix + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
