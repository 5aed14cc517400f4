#!/bin/bash
# prof_round3.sh TAG - the round's evidence run on the GPU box (via gpurun): the full bench line, kernel statistics of the same
# command under rocprofv3 (C2 and T1), the roctx phase ranges (--marker-trace), MFMA utilisation and HBM traffic from separate PMC
# passes.  Results under gpurun_out/round_TAG/ (copy what is judged into profiles/).
tag=${1:-x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/round_$tag
mkdir -p $out
cd $R
python3 bench.py --steps 20 --warmup 5 > $out/bench_c2.json 2> $out/bench_c2.err
cut -c1-260 $out/bench_c2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o c2 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras > $out/bench_c2_prof.json 2> $out/bench_c2_prof.err
f=$(ls $out/prof_c2/*kernel_stats.csv $out/prof_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_t1 -o t1 -- python3 $R/bench.py --n 1000 --m 2000 --steps 2 --warmup 1 --no-cpu --no-extras > $out/bench_t1.json 2> $out/bench_t1.err
f=$(ls $out/prof_t1/*kernel_stats.csv $out/prof_t1/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/t1_kernel_stats.csv
cut -c1-200 $out/bench_t1.json
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $out/prof_mk -o mk -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extras > $out/bench_mk.json 2> $out/bench_mk.err
for f in $(ls $out/prof_mk/*marker*stats*.csv $out/prof_mk/*/*marker*stats*.csv 2>/dev/null); do cp $f $out/c2_marker_stats.csv; done
ls $out/prof_mk $out/prof_mk/* 2>/dev/null | head -20
rm -rf $out/prof_c2 $out/prof_t1 $out/prof_mk
cd $R
bash tools/pmc_mfma.sh r3c2 > $out/pmc_mfma_c2.txt 2>&1
bash tools/pmc_mfma.sh r3t1 --n 1000 --m 2000 > $out/pmc_mfma_t1.txt 2>&1
bash tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1
python3 tools/pmc_traffic_table.py $R/gpurun_out > $out/pmc_traffic_c2.txt 2>&1
cat $out/pmc_mfma_c2.txt | tail -12; cat $out/pmc_traffic_c2.txt | head -8
find $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_mfma_r3c2 $R/gpurun_out/pmc_mfma_r3t1 -name "*.csv" -size +4M -delete 2>/dev/null
