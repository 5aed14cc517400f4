#!/bin/bash
# developer tool (GPU box, via gpurun; round 6): the one-launch kernel of commit f33223b - the form whose -DS1_DEBUG build
# walked other iterates than its release build on 5 of 400 shapes - built four times on the build host (scratch_f33/libs_{rel,dbg,rel_nc,dbg_nc}:
# release / debug, each with and without -ffp-contract=off): if the two builds WITHOUT contraction agree, the difference was contraction
# (fused multiply-adds formed differently in the two code shapes); if they still differ, it is not.
cd $GRAFT_REPO_ROOT
first=${1:-80000}; count=${2:-400}
for v in rel dbg rel_nc dbg_nc; do
  HIPSDP_LIB=$PWD/scratch_f33/libs_$v/libhipsdp.so python3 tests/devtools/solve1_dump.py $first $count gpurun_out/f33_dump_$v.txt || exit 1
done
for pair in "rel dbg" "rel_nc dbg_nc" "rel rel_nc"; do
  set -- $pair
  n=$(diff gpurun_out/f33_dump_$1.txt gpurun_out/f33_dump_$2.txt | grep -c '^<')
  echo "$1 against $2: $n of $count shapes differ: $(diff gpurun_out/f33_dump_$1.txt gpurun_out/f33_dump_$2.txt | grep '^<' | cut -d' ' -f2 | tr '\n' ' ')"
done
