#!/bin/bash
# second half of solve1_contract_ab.sh: which part of the debug build makes the difference - the full fences of S1_WSYNC or the rest
# (poisoned memory, uniformity checks, i.e. another code shape)?  dbg_relsync_nc = debug build with the RELEASE form of S1_WSYNC,
# rel_fullsync_nc = release build with the DEBUG form of S1_WSYNC; everything without contraction.
cd $GRAFT_REPO_ROOT
first=${1:-80150}; count=${2:-250}
for v in rel_nc dbg_nc dbg_relsync_nc rel_fullsync_nc; do
  HIPSDP_LIB=$PWD/scratch_f33/libs_$v/libhipsdp.so python3 tests/devtools/solve1_dump.py $first $count gpurun_out/f33b_dump_$v.txt > /dev/null || exit 1
done
for pair in "rel_nc dbg_nc" "rel_nc rel_fullsync_nc" "dbg_nc dbg_relsync_nc" "rel_nc dbg_relsync_nc" "dbg_nc rel_fullsync_nc"; do
  set -- $pair
  echo "$1 against $2: $(diff gpurun_out/f33b_dump_$1.txt gpurun_out/f33b_dump_$2.txt | grep -c '^<') of $count shapes differ: $(diff gpurun_out/f33b_dump_$1.txt gpurun_out/f33b_dump_$2.txt | grep '^<' | cut -d' ' -f2 | tr '\n' ' ')"
done
