"""Dev helper: print a window of the kernel timeline (start, duration, gap to the previous kernel's
end, queue) from a rocprofv3 --kernel-trace .db.   python tools/timeline.py <results.db> [first] [count]"""
import sqlite3
import sys


def main(db, first=-60, count=40):
    c = sqlite3.connect(db)
    cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = c.execute(f"select name, start, end, {q} from kernels order by start").fetchall()
    first = int(first)
    first = len(rows) + first if first < 0 else first
    t0 = rows[first][1]
    prev_end = None
    for name, s, e, qid in rows[first:first + int(count)]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q{qid}  {name[:70]}")
        prev_end = max(prev_end or e, e)


if __name__ == "__main__":
    main(*sys.argv[1:4])
