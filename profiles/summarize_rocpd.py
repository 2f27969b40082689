#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite database (--kernel-trace --stats) into a small text summary that
can be committed under profiles/.  Usage: python profiles/summarize_rocpd.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
        "max(grid_x*grid_y*grid_z), max(workgroup_x*workgroup_y*workgroup_z) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % db,
             "%-72s %6s %12s %12s %12s %12s %6s %5s %5s %5s %8s %10s %5s" %
             ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds_B", "grid", "wg")]
    for r in rows:
        name = r[0] if len(r[0]) <= 72 else r[0][:69] + "..."
        lines.append("%-72s %6d %12.1f %12.2f %12.2f %12.2f %6.2f %5d %5d %5d %8d %10d %5d" %
                     (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
                      r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
