#!/bin/bash
# developer tool: kernel statistics (rocprofv3) of any python script of the repository: prof_any.sh SCRIPT [ARGS...]
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_any -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_any.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_any/p_kernel_stats.csv gpurun_out/bnb_kernel_stats.csv
python3 tools/show_kernel_stats.py 2>/dev/null | head -${TOP:-30}
tail -3 gpurun_out/prof_any.txt
rm -rf gpurun_out/prof_any
