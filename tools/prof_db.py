#!/usr/bin/env python3
"""prof_db.py RESULTS.db [ITERATIONS] - per-kernel totals from a rocprofv3 rocpd database (ROCm 7.2 writes SQLite by default)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
it = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
print("total kernel ms %.1f, launches %d; per iteration: %.1f launches, %.1f us" % (tot / 1e6, n, n / it, tot / it / 1e3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%8.2f ms %7d calls (%6.1f/iter) %8.1f us  %s" % (r[2] / 1e6, r[1], r[1] / it, r[3] / 1e3, r[0][:90]))
