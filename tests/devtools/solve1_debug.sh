#!/bin/bash
# developer tool (run on the GPU box through gpurun): the one-launch kernel in its debug build (csrc/solve1_body.h, -DS1_DEBUG: LDS and
# workspace start as NaN, wave-level synchronisations are full fences, values declared wave-uniform are checked) against the release
# build on the shapes of solve1_fuzz.py: every line of the two dumps must be the same.  The debug library is built HERE from the
# shipped sources (two minutes) into a copy of the package.
# usage: bash tests/devtools/solve1_debug.sh [first seed] [count]
first=${1:-30000}; count=${2:-200}
R=$GRAFT_REPO_ROOT
cd $R
python3 tests/devtools/solve1_dump.py $first $count gpurun_out/s1_dump_release.txt || exit 1
rm -rf /tmp/dbg && mkdir -p /tmp/dbg && cp -r scip-sdp_amd include /tmp/dbg/ && rm -rf /tmp/dbg/scip-sdp_amd/build /tmp/dbg/scip-sdp_amd/lib
make -C /tmp/dbg/scip-sdp_amd -j16 EXTRA=-DS1_DEBUG > gpurun_out/s1_debug_build.log 2>&1 || { tail gpurun_out/s1_debug_build.log; exit 1; }
# the debug libraries are loaded from /tmp/dbg through HIPSDP_LIB (binding.py): the shipped scip-sdp_amd/lib/*.so are never overwritten
HIPSDP_LIB=/tmp/dbg/scip-sdp_amd/lib/libhipsdp.so python3 tests/devtools/solve1_dump.py $first $count gpurun_out/s1_dump_debug.txt || exit 1
if diff gpurun_out/s1_dump_release.txt gpurun_out/s1_dump_debug.txt > gpurun_out/s1_dump_diff.txt; then echo "debug build = release build on $count shapes from seed $first"; else echo "DIFFERENCES:"; head -20 gpurun_out/s1_dump_diff.txt; fi
