#!/bin/bash
# developer tool (GPU box, via gpurun): the one-launch kernel of the CURRENT build against the libraries kept in scip-sdp_amd/lib_old/
# (copied there by hand before an edit): wave profile of example_TT on both, and the dump of N fuzz shapes compared bit for bit.
# usage: bash tests/devtools/solve1_ab.sh [first seed] [count]
first=${1:-30000}; count=${2:-150}
cd $GRAFT_REPO_ROOT
echo "== new build"; python3 tests/devtools/solve1_waves.py > gpurun_out/ab_waves_new.txt 2>&1; tail -4 gpurun_out/ab_waves_new.txt | cut -c1-400
python3 tests/devtools/solve1_dump.py $first $count gpurun_out/ab_dump_new.txt || exit 1
if [ -f scip-sdp_amd/lib_old/libhipsdp.so ]; then
  echo "== old build"; HIPSDP_LIB=$PWD/scip-sdp_amd/lib_old/libhipsdp.so python3 tests/devtools/solve1_waves.py > gpurun_out/ab_waves_old.txt 2>&1; tail -4 gpurun_out/ab_waves_old.txt | cut -c1-400
  HIPSDP_LIB=$PWD/scip-sdp_amd/lib_old/libhipsdp.so python3 tests/devtools/solve1_dump.py $first $count gpurun_out/ab_dump_old.txt || exit 1
  if diff gpurun_out/ab_dump_old.txt gpurun_out/ab_dump_new.txt > gpurun_out/ab_dump_diff.txt; then echo "new build = old build bit for bit on $count shapes from seed $first"; else echo "DIFFERENCES: $(grep -c '^<' gpurun_out/ab_dump_diff.txt) shapes"; head -6 gpurun_out/ab_dump_diff.txt | cut -c1-200; fi
fi
