#!/bin/bash
# developer tool (GPU box, via gpurun): fuzz shapes iteration by iteration (solve1_fuzz_one.py) under the release build and under a
# -DS1_DEBUG build made here.  usage: bash tests/devtools/solve1_debug_one.sh seed [seed ...]
R=$GRAFT_REPO_ROOT
cd $R
echo "== release build"; python3 tests/devtools/solve1_fuzz_one.py "$@" 2>&1 | cut -c1-200
rm -rf /tmp/dbg && mkdir -p /tmp/dbg && cp -r scip-sdp_amd include /tmp/dbg/ && rm -rf /tmp/dbg/scip-sdp_amd/build /tmp/dbg/scip-sdp_amd/lib
make -C /tmp/dbg/scip-sdp_amd -j16 EXTRA="-DS1_DEBUG $S1_EXTRA" > gpurun_out/s1_debug_build.log 2>&1 || { tail gpurun_out/s1_debug_build.log; exit 1; }
# (loaded through HIPSDP_LIB: the shipped libraries stay in place)
echo "== debug build"; HIPSDP_LIB=/tmp/dbg/scip-sdp_amd/lib/libhipsdp.so python3 tests/devtools/solve1_fuzz_one.py "$@" 2>&1 | cut -c1-200
