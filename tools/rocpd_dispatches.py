#!/usr/bin/env python3
"""One PMC counter of one kernel, dispatch by dispatch, in dispatch order (rocprofv3 rocpd sqlite output): what a sweep
that runs the same kernel under different conditions needs -- the per-kernel sums of rocpd_summary.py average them away.
usage: rocpd_dispatches.py <results.db> <kernel name substring> <counter>"""
import sqlite3
import sys


def main():
    path, pat, counter = sys.argv[1], sys.argv[2], sys.argv[3]
    c = sqlite3.connect(path)
    cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
    rows = c.execute("select %s, %s, sum(value) from counters_collection where counter_name = ? and %s like ? group by 2 order by 2"
                     % (namecol, order, namecol), (counter, "%" + pat + "%")).fetchall()
    print("%-8s %-20s %s" % ("dispatch", counter, "kernel"))
    for i, (name, disp, v) in enumerate(rows):
        print("%-8d %-20.0f %s" % (i, v, name[:90]))


if __name__ == "__main__":
    main()
