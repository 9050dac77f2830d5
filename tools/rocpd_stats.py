#!/usr/bin/env python3
"""Per-kernel statistics (calls, total, average, share) from a rocprofv3 rocpd SQLite result (`*_results.db`), printed as CSV:
what `rocprofv3 --kernel-trace --stats` reports, for runs whose output format was the default database.
    python tools/rocpd_stats.py gpurun_out/prof/p_results.db [> profiles/rNN_kernel_stats.csv]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = list(db.execute(q))
    total = float(sum(r[2] for r in rows)) or 1.0
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, t, a, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n.replace('"', "'"), c, t, a, 100.0 * t / total, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1])
