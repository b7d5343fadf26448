#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (sqlite) outputs: per-kernel time stats and PMC counter sums.
usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        print("==", path)
        try:
            rows = c.execute(
                "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                "from kernels group by name order by 3 desc").fetchall()
            tot = sum(r[2] for r in rows) or 1
            print("%-72s %6s %14s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
            for r in rows:
                print("%-72s %6d %14d %12.0f %12d %12d %6.2f" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
        except sqlite3.Error as e:
            print("no kernel table:", e)
        try:
            cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
            namecol = "kernel_name" if "kernel_name" in cols else "name"
            rows = c.execute(
                "select %s, counter_name, count(*), sum(value), avg(value) from counters_collection "
                "group by 1, 2 order by 1, 2" % namecol).fetchall()
            if rows:
                print("%-60s %-24s %6s %20s %18s" % ("kernel", "counter", "n", "sum", "avg/dispatch"))
            for r in rows:
                print("%-60s %-24s %6d %20.0f %18.1f" % (r[0][:60], r[1], r[2], r[3], r[4]))
        except sqlite3.Error as e:
            print("no counters:", e)


if __name__ == "__main__":
    main()
