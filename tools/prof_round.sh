#!/bin/bash
# prof_round.sh TAG - the round's evidence run on the GPU box (via gpurun): bench line with cpu_baseline, kernel statistics of the
# same command under rocprofv3 (CSV), and the T1 line; results under gpurun_out/round_TAG/ (copy what is judged into profiles/)
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/round_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 1 > $out/bench_c2.json 2> $out/bench_c2.err
cut -c1-300 $out/bench_c2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $out/bench_c2_prof.json 2> $out/bench_c2_prof.err
f=$(ls $out/prof_c2/*kernel_stats.csv $out/prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv && head -5 $f | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_t1 -o t1 -- python3 $GRAFT_REPO_ROOT/bench.py --n 1000 --m 2000 --steps 2 --warmup 1 --no-cpu > $out/bench_t1.json 2> $out/bench_t1.err
f=$(ls $out/prof_t1/*kernel_stats.csv $out/prof_t1/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/t1_kernel_stats.csv
cut -c1-300 $out/bench_t1.json
rm -rf $out/prof_c2 $out/prof_t1
