#!/bin/bash
# solve1_pmc.sh TAG - PMC summary of the one-launch kernel on the example_TT tree (the part of tools/prof_round5.sh that can run alone)
tag=${1:-x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/round_$tag
mkdir -p $out
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
cat $out/solve1_pmc.txt
rm -rf $out/s1_SQ_*
