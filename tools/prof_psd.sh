#!/bin/bash
# prof_psd.sh - developer tool: kernel and HIP API statistics of the fused PSD projection at n = 200 (tests/devtools/psd_project_time.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_psd
rocprofv3 --kernel-trace --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/prof_psd -o p -- python3 $R/tests/devtools/psd_project_time.py > $R/gpurun_out/prof_psd.txt 2>&1
tail -2 $R/gpurun_out/prof_psd.txt
f=$(find $R/gpurun_out/prof_psd -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-120
f=$(find $R/gpurun_out/prof_psd -name "*hip_api_stats.csv" | head -1); head -10 $f | cut -c1-120
rm -rf $R/gpurun_out/prof_psd
