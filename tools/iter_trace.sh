#!/bin/bash
# iter_trace.sh TAG - kernel trace of a short C2 bench run and the timeline of its iterations (tools/iter_timeline.py)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/itrace_$tag -o k -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-extras "$@" > /dev/null 2>&1 || exit 1
f=$(find $R/gpurun_out/itrace_$tag -name "*kernel_trace.csv" | head -1)
python3 $R/tools/iter_timeline.py $f > $R/gpurun_out/itrace_$tag.txt
rm -rf $R/gpurun_out/itrace_$tag
cat $R/gpurun_out/itrace_$tag.txt
