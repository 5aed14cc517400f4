#!/bin/bash
# iter_seq.sh TAG - kernel trace of a short C2 bench run: the per-kernel table of iterations (tools/iter_timeline.py) and one iteration
# kernel by kernel with the idle time in front of every kernel (tools/iter_sequence.py)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/itrace_$tag -o k -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-extras "$@" > /dev/null 2>&1 || exit 1
f=$(find $R/gpurun_out/itrace_$tag -name "*kernel_trace.csv" | head -1)
python3 $R/tools/iter_timeline.py $f > $R/gpurun_out/itrace_$tag.txt
python3 $R/tools/iter_sequence.py $f 3 > $R/gpurun_out/iseq_$tag.txt
rm -rf $R/gpurun_out/itrace_$tag
head -8 $R/gpurun_out/itrace_$tag.txt; cat $R/gpurun_out/iseq_$tag.txt
