#!/usr/bin/env python
"""Timeline of the LAST pass of a profiled program from a rocprofv3 rocpd database (--kernel-trace): every kernel launched
after the last launch whose name contains `marker`, in start order, with its duration and the gap to the previous kernel's
end - where a short multi-launch sequence (a rank of the sharded build: ~5 ms, ~90 launches) spends its time, idle gaps
(host read-backs) included.  usage: rocpd_timeline.py results.db marker [max_rows]"""
import sqlite3
import sys


def main(path, marker, max_rows=400):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    s_col = "start" if "start" in cols else "start_time"
    e_col = "end" if "end" in cols else "end_time"
    rows = cur.execute("select name, %s, %s from kernels order by %s" % (s_col, e_col, s_col)).fetchall()
    last = max((i for i, r in enumerate(rows) if marker in r[0]), default=0)
    rows = rows[last:]
    t0 = rows[0][1]
    prev_end = t0
    busy = 0.0
    print("# %d launches after the last '%s'; times in microseconds from its start" % (len(rows), marker))
    print("%10s %9s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
    agg = {}
    for name, s, e in rows[:max_rows]:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short.split("(")[0][:90] or name[:90]
        print("%10.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
        busy += (e - s) / 1e3
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print("# span %.1f us, kernels busy %.1f us (%.0f %%), idle %.1f us" % (span, busy, 100 * busy / max(span, 1e-9), span - busy))
    print("# by kernel:")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("#   %9.1f us  x%-3d %s" % (t, c, k))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 400)
