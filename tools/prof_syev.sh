#!/bin/bash
# prof_syev.sh n... - developer tool: kernel statistics of the device eigendecomposition (block Jacobi above 128 rows) at the given sizes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/tests/devtools/syev_time.py "$@"
for n in "$@"; do
  rm -rf $R/gpurun_out/prof_syev
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_syev -o s -- python3 $R/tests/devtools/syev_time.py $n > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_syev -name "*kernel_stats.csv" | head -1)
  echo "== n = $n (5 decompositions)"; head -8 $f | cut -c1-140
done
rm -rf $R/gpurun_out/prof_syev
