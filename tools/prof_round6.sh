#!/bin/bash
# prof_round6.sh TAG - (round 6: prof_round5.sh with the iteration sequence added and progress lines) the round's evidence run on the GPU box (via gpurun): the full bench line, kernel statistics of the same command
# under rocprofv3 (C2 and T1), MFMA utilisation and L2-miss traffic of the assembly kernels from separate PMC passes, the PMC summary of
# the one-launch kernel on the example_TT tree, the timeline between two assemblies.  Results under gpurun_out/round_TAG/ (what is
# judged is copied into profiles/).
tag=${1:-x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/round_$tag
mkdir -p $out
cd $R
python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
cut -c1-200 $out/bench_c2.json; echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o c2 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras > $out/bench_c2_prof.json 2> $out/bench_c2_prof.err
f=$(find $out/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/c2_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_t1 -o t1 -- python3 $R/bench.py --n 1000 --m 2000 --steps 2 --warmup 1 --no-cpu --no-extras > $out/bench_t1.json 2> $out/bench_t1.err
f=$(find $out/prof_t1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/t1_kernel_stats.csv
rm -rf $out/prof_c2 $out/prof_t1; echo "kernel stats done"
cd $R
bash tools/pmc_mfma.sh r6c2 > $out/pmc_mfma_c2.txt 2>&1
bash tools/pmc_mfma.sh r6t1 --n 1000 --m 2000 > $out/pmc_mfma_t1.txt 2>&1
bash tools/pmc_traffic.sh > $out/pmc_traffic.log 2>&1
python3 tools/pmc_traffic_table.py $R/gpurun_out > $out/pmc_traffic_c2.txt 2>&1; echo "pmc done"
bash tools/iter_seq.sh r6final > $out/iter_sequence_log.txt 2>&1
cp $R/gpurun_out/itrace_r6final.txt $out/iter_timeline.txt; cp $R/gpurun_out/iseq_r6final.txt $out/iter_sequence.txt
echo "timeline done"
# the one-launch kernel on the whole example_TT tree: instruction mix, LDS bank conflicts, waits (one PMC pass per group)
cd /tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  g=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/s1_$g -o p -- python3 $R/tests/devtools/bnb_rate.py TT > /dev/null 2> $out/s1_$g.err
done
python3 - $out <<'PY' > $out/solve1_pmc.txt 2>&1
import csv, collections, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(out + "/s1_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_solve1" not in k: continue
        k = k[k.index("k_solve1"):].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, f)].add(r["Dispatch_Id"])
print("== one-launch kernel on the example_TT tree (tests/devtools/bnb_rate.py TT: cold and warm tree), PMC sums over all launches; three passes")
for k in agg:
    n = max(len(v) for (kk, f), v in calls.items() if kk == k)
    c = agg[k]
    print("%s: %d launches" % (k, n))
    wc = max(c.get("SQ_WAVE_CYCLES", 0), 1.0)
    for nm in sorted(c):
        print("   %-28s %16.0f   per launch %12.0f" % (nm, c[nm], c[nm] / n))
    print("   share of wave cycles: VALU active %.1f %%, LDS active %.1f %%, scalar active %.1f %%, waiting for an instruction %.1f %% (for LDS %.1f %%)" %
          (100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_SCA", 0) / wc,
           100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc))
    if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        print("   LDS bank conflict cycles / LDS active cycles: %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
PY
cat $out/solve1_pmc.txt | head -50
tail -9 $out/pmc_mfma_c2.txt; tail -7 $out/pmc_mfma_t1.txt; head -8 $out/pmc_traffic_c2.txt
find $R/gpurun_out -name "*.csv" -size +2M -delete 2>/dev/null
rm -rf $out/s1_SQ_*
