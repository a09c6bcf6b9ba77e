#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) as a per-kernel table:
calls, total / average / min / max duration, VGPR, LDS.  Usage: rocpd_summary.py results.db > summary.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-70s %7s %14s %14s %12s %12s %6s %5s %8s %7s %9s" % (
        "kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct", "vgpr", "lds_B", "scratch", "grid_x"))
    for r in rows:
        name = r[0]
        if len(name) > 69:
            name = name[:66] + "..."
        print("%-70s %7d %14.3f %14.4f %12.4f %12.4f %6.2f %5d %8d %7d %9d" % (
            name, r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / total, (r[6] or 0) + (r[7] or 0),
            r[8] or 0, r[9] or 0, r[10] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
