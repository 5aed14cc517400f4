#!/bin/bash
# developer script: instruction-cache and wait counters of the one-launch solve (csrc/solve1.hip) on one instance
# usage (on the GPU box): bash tools/pmc_s1.sh <filter of tests/devtools/solve1_dbg.py> <tag>
FLT=${1:-TT}; TAG=${2:-s1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQC_TC_INST_REQ SQC_TC_STALL SQ_INSTS_SALU SQ_INSTS_LDS"; do
   n=$(echo $set | tr ' ' '_' | cut -c 1-40)
   rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$n -o r -- python3 $REPO/tests/devtools/solve1_dbg.py $FLT > $OUT/$n.log 2>&1
done
python3 - <<PY
import glob, csv, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if 'k_solve1' in row.get('Kernel_Name', ''):
            acc[row['Counter_Name']] += float(row['Counter_Value'])
    print(f.split('/')[-2], dict(acc))
PY
