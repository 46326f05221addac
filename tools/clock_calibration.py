"""Match bench.py's per-launch device-clock spans of K1 against the rocprofv3 --kernel-trace durations of the SAME
launches.  Run bench.py under rocprofv3 with MPPI_BENCH_DUMP_LAUNCHES=<json> set; then
    python tools/clock_calibration.py <results.db> <launches.json> <kernel name pattern> [out.txt]
prints, per launch of the timed region, device span vs rocprofv3 duration, and the statistics of the difference --
the dispatch offset bench.py adds (DISPATCH_OFFSET_US)."""
import json
import sqlite3
import sys


def main(db, dump, pattern, out=None):
    d = json.load(open(dump))
    dev = d["k1_device_span_us"]
    n0, n = d["k1_launches_before_timed_region"], len(dev)
    c = sqlite3.connect(db)
    roc = [x / 1e3 for x, in c.execute("select duration from kernels where name like ? order by start", (f"%{pattern}%",))]
    sel = roc[n0:n0 + n]
    lines = [f"# {db}: {len(roc)} dispatches of *{pattern}*; timed region = dispatches [{n0}, {n0 + n}) (after the probe + warm-up launches)",
             "# launch  device_span_us  rocprofv3_us  difference_us"]
    diffs = []
    for i, (a, b) in enumerate(zip(dev, sel)):
        diffs.append(b - a)
        lines.append(f"{i:6d}  {a:14.3f}  {b:12.3f}  {b - a:13.3f}")
    if diffs:
        s = sorted(diffs)
        avg_dev, avg_roc = sum(dev[:len(sel)]) / len(sel), sum(sel) / len(sel)
        lines.append(f"# device span avg {avg_dev:.3f} us | rocprofv3 avg {avg_roc:.3f} us (median {sorted(sel)[len(sel) // 2]:.3f}) | "
                     f"difference avg {sum(diffs) / len(diffs):.3f} median {s[len(s) // 2]:.3f} min {s[0]:.3f} max {s[-1]:.3f} us")
        lines.append(f"# => rocprofv3 = device span + {sum(diffs) / len(diffs):.2f} us on these launches ({(avg_roc / avg_dev - 1) * 100:.2f} %)")
    cold = d.get("k1_cold_device_span_us")
    if cold:
        # the HBM-cold pass of bench.py (k1_hbm_cold): the LAST len(cold) dispatches of the kernel in the trace
        selc = roc[-len(cold):]
        dc = sorted(b - a for a, b in zip(cold, selc))
        med = lambda v: sorted(v)[len(v) // 2]
        lines.append(f"# HBM-cold launches (last {len(cold)} dispatches): device span median {med(cold):.3f} avg {sum(cold) / len(cold):.3f} us | rocprofv3 median "
                     f"{med(selc):.3f} avg {sum(selc) / len(selc):.3f} us | difference avg {sum(dc) / len(dc):.3f} median {dc[len(dc) // 2]:.3f} min {dc[0]:.3f} max {dc[-1]:.3f} us")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print("\n".join(lines[-4:]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
