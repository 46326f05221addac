"""gpurun_out/parity_margins.json (written by the -m gpu tests, tests/margins.py) -> the committed record: the criterion and
the worst margin per (test, quantity), one entry per line.
    python tools/margins_compact.py gpurun_out/parity_margins.json profiles/r03_parity_margins.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
worst = d["worst_per_test_and_quantity"]
with open(dst, "w") as f:
    f.write('{"criterion": %s,\n "entries_recorded": %d,\n "worst_per_test_and_quantity": [\n' % (json.dumps(d["criterion"]), len(d["all"])))
    rows = []
    for k in sorted(worst):
        r = worst[k]
        rows.append("  " + json.dumps({"test": r["test"], "quantity": r["quantity"], "err_over_scale": r["err_over_scale"],
                                       "floor_over_scale": r["floor_over_scale"], "rtol": r["rtol"], **({"note": r["note"]} if r.get("note") else {})}))
    f.write(",\n".join(rows))
    f.write("\n ]}\n")
print(len(rows), "entries ->", dst)
