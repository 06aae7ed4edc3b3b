#!/usr/bin/env python
"""Kernel-time summary from a rocprofv3 rocpd (sqlite) database: name, calls, total/avg/min/max us, %.
usage: python tools/rocpd_stats.py <results.db> [> profiles/xxx_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    q = ("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d "
         "join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("%-90s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, mn, mx in rows:
        print("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name[:90], n, tot / 1e3, tot / 1e3 / n, mn / 1e3, mx / 1e3,
                                                                 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
