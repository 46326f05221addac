"""Where a command's wall time goes BETWEEN its kernels (VERDICT r03 weak #3c: 98.6 us per command against 83.5 + 4.8 us of
kernels on the driver's box).  From the rocprofv3 --kernel-trace sqlite of a bench.py run:
    python tools/timeline_gaps.py <results.db> <first kernel pattern> [<n commands from the end of the run of that kernel>] [out.txt]
finds the dispatches of the command's first kernel (e.g. rollout_onchip_kernel), treats everything from one such dispatch
up to the next as one command, and prints per command: the kernels it ran (name, start offset, duration), the idle gaps
between them, and the period (start to next start); then the averages over the longest run of back-to-back commands
(period < 3x median: the timed region and its warm-up, not the probes around it)."""
import sqlite3
import sys


def main(db, pattern, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    firsts = [i for i, r in enumerate(rows) if pattern in r[0]]
    if len(firsts) < 3:
        print(f"fewer than three dispatches of *{pattern}*")
        return
    cmds = []
    for a, b in zip(firsts[:-1], firsts[1:]):
        ks = rows[a:b]
        period = (rows[b][1] - rows[a][1]) / 1e3
        busy = sum(k[2] - k[1] for k in ks) / 1e3
        gaps = [(ks[i + 1][1] - ks[i][2]) / 1e3 for i in range(len(ks) - 1)] + [(rows[b][1] - ks[-1][2]) / 1e3]
        cmds.append((period, busy, gaps, ks))
    periods = sorted(p for p, *_ in cmds)
    med = periods[len(periods) // 2]
    # longest run of consecutive commands whose period stays below 3x the median
    best, cur = (0, 0), None
    for i, (p, *_r) in enumerate(cmds + [(1e18,)]):
        if p < 3 * med:
            cur = (cur[0], i + 1) if cur else (i, i + 1)
            if cur[1] - cur[0] > best[1] - best[0]:
                best = cur
        else:
            cur = None
    sel = cmds[best[0]:best[1]]
    lines = [f"# {db}: {len(cmds)} commands starting with *{pattern}*; back-to-back run = commands [{best[0]}, {best[1]}) (median period {med:.2f} us)"]
    names = [k[0].split("(")[0][-60:] for k in sel[0][3]]
    n = len(sel)
    same = [s for s in sel if len(s[3]) == len(names)]
    lines.append(f"# per command over that run ({len(same)} of {n} commands with the same {len(names)} kernels):")
    lines.append(f"#   period (start -> next start)   avg {sum(s[0] for s in same) / len(same):8.2f} us   min {min(s[0] for s in same):8.2f}   max {max(s[0] for s in same):8.2f}")
    lines.append(f"#   kernels busy                   avg {sum(s[1] for s in same) / len(same):8.2f} us")
    for i, nm in enumerate(names):
        d = [(s[3][i][2] - s[3][i][1]) / 1e3 for s in same]
        g = [s[2][i] for s in same]
        lines.append(f"#   kernel {i}: {nm:<60} avg {sum(d) / len(d):8.2f} us   then idle avg {sum(g) / len(g):6.2f} us (min {min(g):.2f}, max {max(g):.2f})")
    tot_gap = sum(sum(s[2]) for s in same) / len(same)
    lines.append(f"#   idle between kernels per command: {tot_gap:.2f} us  (= period - busy)")
    lines.append("# command  period_us  busy_us  gaps_us...")
    for i, (p, b, g, ks) in enumerate(sel[:64]):
        lines.append(f"{best[0] + i:6d}  {p:9.2f}  {b:8.2f}  " + "  ".join(f"{x:6.2f}" for x in g))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print("\n".join(l for l in lines if l.startswith("# ")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
