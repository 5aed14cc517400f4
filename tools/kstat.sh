#!/bin/bash
# kstat.sh TAG [bench args] - kernel statistics (rocprofv3 --kernel-trace --stats) of a short C2 bench run on the GPU box; the
# statistics file lands in gpurun_out/kstat_TAG.csv.  HIPSDP_LIB etc. are inherited.
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstat_$tag -o k -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras "$@" > /dev/null 2>&1 || exit 1
f=$(find $R/gpurun_out/kstat_$tag -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/kstat_$tag.csv
rm -rf $R/gpurun_out/kstat_$tag
