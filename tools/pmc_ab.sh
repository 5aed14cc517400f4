#!/bin/bash
# pmc_ab.sh - HBM traffic and MFMA utilisation of the Schur kernels for the variants of the assembly (environment switches), C2.
# usage (on the GPU box): bash tools/pmc_ab.sh ; tables in gpurun_out/pmc_ab_*.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for v in "HIPSDP_GEMM4=0" "HIPSDP_GEMM4=1" "HIPSDP_SCHUR_LEFT=1"; do
  i=$((i+1))
  export $v
  echo "== variant $i: $v"
  rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
  bash $R/tools/pmc_traffic.sh > /dev/null 2>&1
  python3 $R/tools/pmc_traffic_table.py $R/gpurun_out > $R/gpurun_out/pmc_ab_traffic_$i.txt 2>&1
  head -12 $R/gpurun_out/pmc_ab_traffic_$i.txt
  bash $R/tools/pmc_mfma.sh ab$i > $R/gpurun_out/pmc_ab_mfma_$i.txt 2>&1
  tail -10 $R/gpurun_out/pmc_ab_mfma_$i.txt
  unset ${v%%=*}
  find $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_mfma_ab$i -name "*.csv" -size +2M -delete 2>/dev/null
done
