"""Match bench.py's per-launch device-clock spans against the rocprofv3 --kernel-trace durations of the SAME launches
(VERDICT r02 item 2).  Run bench.py under rocprofv3 with MPPI_BENCH_DUMP_LAUNCHES=<json> set; then
    python tools/clock_calibration.py <results.db> <launches.json> [out.txt]
prints, per region bench.py dumped (headline / streaming / hbm_cold: kernel name pattern, spans, how many dispatches of
that kernel precede the region), device span vs rocprofv3 duration per launch and the statistics of the difference --
the dispatch offsets bench.py adds (DISPATCH_OFFSET_US_*)."""
import json
import sqlite3
import sys


def main(db, dump, out=None):
    d = json.load(open(dump))
    c = sqlite3.connect(db)
    lines = []
    med = lambda v: sorted(v)[len(v) // 2]
    for name, reg in d["regions"].items():
        dev, pat, n0 = reg["spans_us"], reg["pattern"], reg["launches_before"]
        roc = [x / 1e3 for x, in c.execute("select duration from kernels where name like ? order by start", (f"%{pat}%",))]
        if n0 < 0:
            # position not known to bench.py (other launches of the kernel follow): the run of len(dev) dispatches whose
            # durations track the device spans best (smallest spread of the differences)
            best = None
            for o in range(len(roc) - len(dev) + 1):
                df = [roc[o + i] - dev[i] for i in range(len(dev))]
                m = sum(df) / len(df)
                v = sum((x - m) ** 2 for x in df)
                if best is None or v < best[0]:
                    best = (v, o)
            n0 = best[1]
            where = f"dispatches [{n0}, {n0 + len(dev)}) (located by matching the spans)"
        else:
            where = f"dispatches [{n0}, {n0 + len(dev)})"
        sel = roc[n0:n0 + len(dev)]
        lines.append(f"# region {name}: {len(roc)} dispatches of *{pat}* in {db}; this region = {where}")
        lines.append("# launch  device_span_us  rocprofv3_us  difference_us")
        diffs = []
        for i, (a, b) in enumerate(zip(dev, sel)):
            diffs.append(b - a)
            lines.append(f"{i:6d}  {a:14.3f}  {b:12.3f}  {b - a:13.3f}")
        if diffs:
            n = len(diffs)
            lines.append(f"# {name}: device span avg {sum(dev[:n]) / n:.3f} median {med(dev[:n]):.3f} us | rocprofv3 avg {sum(sel) / n:.3f} median "
                         f"{med(sel):.3f} us | difference avg {sum(diffs) / n:.3f} median {med(diffs):.3f} min {min(diffs):.3f} max {max(diffs):.3f} us "
                         f"=> rocprofv3 = device span + {sum(diffs) / n:.2f} us ({(sum(sel) / sum(dev[:n]) - 1) * 100:.2f} %)")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print("\n".join(l for l in lines if l.startswith("# ") and "=>" in l))


if __name__ == "__main__":
    main(*sys.argv[1:4])
