#!/usr/bin/env python3
"""iter_sequence.py KERNEL_TRACE_CSV [iteration-from-the-end] - developer tool: ONE iteration of the bench solve, kernel by kernel, between the end
of one Schur assembly and the start of the next: start (us after the assembly's end), duration, the time before it in which NO kernel was running,
queue (stream) and name - where the device idles and what it waits for."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gram = [i for i, r in enumerate(rows) if r[2].startswith("hs_gram_kernel") or r[2].startswith("hs_dgemm2_kernel<0, 0>")]
first = [i for i, r in enumerate(rows) if r[2].startswith("hs_dgemm5_kernel<1, 1>") or r[2].startswith("hs_dgemm2_kernel<1, 1>")]
spans = []
for g in gram:
    nxt = [f for f in first if f > g]
    if nxt and rows[nxt[0]][0] - rows[g][1] < 6e6:
        spans.append((g, nxt[0]))
g, f = spans[-back]
t0 = rows[g][1]
cur = t0
idle = 0.0
print("%9s %8s %8s  %-6s %s" % ("start us", "dur us", "idle us", "queue", "kernel"))
for (a, b, nm, qid) in rows[g + 1:f + 1]:
    gap = max(0, a - cur)
    idle += gap
    print("%9.1f %8.1f %8.1f  %-6s %s" % ((a - t0) / 1e3, (b - a) / 1e3, gap / 1e3, qid, nm[:70]))
    cur = max(cur, b)
print("span %.1f us, no kernel running %.1f us" % ((rows[f][0] - t0) / 1e3, idle / 1e3))
