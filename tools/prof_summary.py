"""Summarise a rocprofv3 --kernel-trace sqlite (.db) into a text table (name, calls, avg/min/max us,
share, grid, vgpr, scratch) -- what gets committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), avg(duration), sum(duration), min(duration), max(duration), "
        "max(grid_x), max(grid_y), max(workgroup_x), max(vgpr_count), max(scratch_size), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows) or 1
    med = {}
    for name, in c.execute("select distinct name from kernels"):
        ds = sorted(d for d, in c.execute("select duration from kernels where name = ?", (name,)))
        med[name] = ds[len(ds) // 2]
    lines = [f"# rocprofv3 --kernel-trace summary of {db}",
             f"{'share':>6} {'calls':>6} {'avg_us':>9} {'med_us':>9} {'min_us':>9} {'max_us':>9} {'grid':>14} {'wg':>4} {'vgpr':>4} {'scr':>4} {'lds':>6}  kernel"]
    for r in rows[:40]:
        lines.append(f"{r[3] / tot * 100:5.1f}% {r[1]:6d} {r[2] / 1e3:9.1f} {med[r[0]] / 1e3:9.1f} {r[4] / 1e3:9.1f} {r[5] / 1e3:9.1f} "
                     f"{str(r[6]) + 'x' + str(r[7]):>14} {r[8]:4d} {r[9]:4d} {r[10]:4d} {r[11]:6d}  {r[0][:140]}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
