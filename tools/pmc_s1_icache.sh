#!/bin/bash
# pmc_s1_icache.sh - instruction-cache counters of the one-launch kernel on example_TT's tree (is the kernel waiting for its own instructions?)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/s1_icache
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM"; do
  g=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$g -o p -- python3 $R/tests/devtools/bnb_rate.py TT > /dev/null 2> $out/$g.err || { tail -3 $out/$g.err; exit 1; }
done
python3 - $out <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_solve1" not in k: continue
        k = k[k.index("k_solve1"):].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, f)].add(r["Dispatch_Id"])
for k in agg:
    n = max(len(v) for (kk, f), v in calls.items() if kk == k)
    print("%s: %d launches" % (k, n))
    for nm in sorted(agg[k]):
        print("   %-30s %16.0f   per launch %12.0f" % (nm, agg[k][nm], agg[k][nm] / n))
PY
