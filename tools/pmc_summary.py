"""Summarise PMC counters of a rocprofv3 --pmc run (.db): per kernel name, mean counter value per
dispatch.  Usage: python tools/pmc_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    lines = [f"# rocprofv3 --pmc summary of {db}"]
    view = "counters_collection" if "counters_collection" in names else None
    if view is None:
        lines.append("no counters_collection view; tables: " + ", ".join(n for n in names if "pmc" in n.lower() or "counter" in n.lower()))
    else:
        cols = [d[1] for d in c.execute(f"pragma table_info({view})")]
        lines.append("columns: " + ", ".join(cols))
        kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
        ccol = "counter_name" if "counter_name" in cols else None
        vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        if ccol and vcol:
            q = (f"select {kcol}, {ccol}, count(*), avg({vcol}), min({vcol}), max({vcol}) from {view} "
                 f"group by {kcol}, {ccol} order by avg({vcol}) desc")
            lines.append(f"{'counter':>14} {'dispatches':>10} {'mean':>16} {'min':>16} {'max':>16}  kernel")
            for r in c.execute(q):
                lines.append(f"{r[1]:>14} {r[2]:10d} {r[3]:16.1f} {r[4]:16.1f} {r[5]:16.1f}  {str(r[0])[:120]}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
