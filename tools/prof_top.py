#!/usr/bin/env python3
"""prof_top.py <results.db> [N] - per-kernel totals of a rocprofv3 --kernel-trace database (developer tool)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = list(db.cursor().execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
for r in rows[:n]:
    print("%-100s %6d %10.3f ms %10.1f us %5.1f%%" % (r[0][:100], r[1], r[2] / 1e6, r[3] / 1e3, 100 * r[2] / tot))
print("total kernel time %.3f ms, %d launches" % (tot / 1e6, sum(r[1] for r in rows)))
