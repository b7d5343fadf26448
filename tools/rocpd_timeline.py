#!/usr/bin/env python3
"""The last N kernel dispatches of a rocprofv3 kernel trace (rocpd sqlite) as a timeline: start offset, duration, gap to
the end of the previous dispatch.  usage: rocpd_timeline.py <results.db> [N]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = c.execute("select name, start, end from kernels order by start").fetchall()[-n:]
t0, prev = rows[0][1], None
for name, s, e in rows:
    print("%10.1f us  dur %9.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, name[:70]))
    prev = e
