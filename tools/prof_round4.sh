#!/bin/bash
# prof_round4.sh TAG - the round's evidence run on the GPU box (via gpurun): the full bench line, kernel statistics of the same
# command under rocprofv3 (C2), kernel statistics of the B&B trees (k_solve1: one launch per node solve), the per-wavefront phase
# profile of the one-launch solve, the stage times of the solver interface.  Results under gpurun_out/round_TAG/ (copy what is
# judged into profiles/).  The Schur kernels did not change this round: their PMC profiles are profiles/r03_e_pmc_*.
tag=${1:-x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/round_$tag
mkdir -p $out
cd $R
python3 bench.py --steps 20 --warmup 5 > $out/bench_c2.json 2> $out/bench_c2.err
cut -c1-260 $out/bench_c2.json
python3 tests/devtools/solve1_waves.py > $out/solve1_wave_profile.txt 2> $out/solve1_wave_profile.err
tail -3 $out/solve1_wave_profile.txt
HIPSDP_STAGE_TIMES=1 python3 tests/devtools/bnb_stage_times.py > $out/stage_times.txt 2>&1
tail -2 $out/stage_times.txt
python3 tests/devtools/solve1_host_overhead.py > $out/solve1_host_overhead.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o c2 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras > $out/bench_c2_prof.json 2> $out/bench_c2_prof.err
f=$(ls $out/prof_c2/*kernel_stats.csv $out/prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bnb -o bnb -- python3 $R/tests/devtools/bnb_rate.py > $out/bnb_rate_prof.txt 2> $out/bnb_rate_prof.err
f=$(ls $out/prof_bnb/*kernel_stats.csv $out/prof_bnb/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/bnb_kernel_stats.csv
rm -rf $out/prof_c2 $out/prof_bnb
cd $R
python3 tests/devtools/bnb_rate.py > $out/bnb_rate.txt 2>&1
cat $out/bnb_rate.txt
head -5 $out/bnb_kernel_stats.csv
python3 tests/devtools/solve1_sizes.py > $out/solve1_sizes.txt 2>&1
cat $out/solve1_sizes.txt
