#!/bin/bash
# ab_gemm_order_kernels.sh - developer tool: average duration of the assembly kernels inside the bench solve (rocprofv3 --kernel-trace --stats)
# with both list orders of the paired-band kernel, alternating, C2 and T1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for ord in 0 1; do
for size in "500 1000 5" "1000 2000 2"; do
  set -- $size
  export HIPSDP_GEMM_ORDER=$ord
  rm -rf $R/gpurun_out/ab_ord
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_ord -o x -- python3 $R/bench.py --n $1 --m $2 --steps $3 --warmup 1 --no-cpu --no-extras > /dev/null 2>&1
  f=$(find $R/gpurun_out/ab_ord -name "*kernel_stats.csv" | head -1)
  python3 - $f $ord $1 <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"]
    if "dgemm5" in k or "gram_kernel" in k or "dgemm2_kernel<0, 0>" in k:
        print("order %s n %4s  %-34s calls %3s  avg %.3f ms" % (sys.argv[2], sys.argv[3], k.split("(")[0].replace("void ", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done; done; done
rm -rf $R/gpurun_out/ab_ord
