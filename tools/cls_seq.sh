#!/bin/bash
# cls_seq.sh - kernel trace of three root solves of example_CLS (the general path: the one-launch kernel declines it) and the kernels of ONE
# interior-point iteration in order (start, duration, idle time in front, queue)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cls_trace -o k -- python3 $R/tests/devtools/solve1_tt.py example_CLS.dat-s.gz > $R/gpurun_out/cls_run.txt 2>&1 || exit 1
f=$(find $R/gpurun_out/cls_trace -name "*kernel_trace.csv" | head -1)
python3 - $f > $R/gpurun_out/cls_seq.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
# an iteration = from one k_schur_small (the assembly of B&B-sized problems) to the next
idx = [i for i, r in enumerate(rows) if "schur_small" in r[2]]
print("assemblies found:", len(idx))
a, b = idx[-6], idx[-5]
t0 = rows[a][0]; cur = t0; idle = 0.0
for (s, e, nm, q) in rows[a:b]:
    gap = max(0, s - cur); idle += gap
    print("%8.1f %7.1f %7.1f  %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, q, nm[:80]))
    cur = max(cur, e)
print("iteration span %.1f us, no kernel running %.1f us, %d kernels" % ((rows[b][0] - t0) / 1e3, idle / 1e3, b - a))
PY
rm -rf $R/gpurun_out/cls_trace
cat $R/gpurun_out/cls_seq.txt; tail -1 $R/gpurun_out/cls_run.txt
