#!/bin/bash
# prof_bnb.sh INSTANCE [MAXNODES] - kernel statistics of a B&B run over the HIP backend (run on the GPU box via gpurun)
inst=${1:-example_CLS.dat-s.gz}; nodes=${2:-100000}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bnb -o b -- python3 $GRAFT_REPO_ROOT/tests/devtools/bnb_gpu_run.py $inst $nodes > $GRAFT_REPO_ROOT/gpurun_out/bnb.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/bnb.log
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_bnb/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_bnb/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/bnb_kernel_stats.csv && head -40 $f | cut -c1-160
