#!/bin/bash
# pmc_gram_l2.sh - developer tool (round 6): L2 hits / misses / requests of the Gram product at the bench shape
# (M = 1001, K = 250000) through the Gram kernel (csrc/gram.hip) and through the K-sliced tile kernel (HIPSDP_GRAM=0), per dispatch and - as
# far as the profiler splits them - per XCD.  Counters in passes of their own, --kernel-trace only (no other trace domain).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gram_one.py <<'PY'
import os, sys, importlib.util
ROOT = os.environ["GRAFT_REPO_ROOT"]
spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
print(hb.gram_selfcheck(1001, 250000, reps=3))
PY
for mode in gram tile; do
  if [ $mode = tile ]; then export HIPSDP_GRAM=0; else unset HIPSDP_GRAM; fi
  for ctr in "TCC_HIT TCC_MISS" "TCC_REQ TCC_READ" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" "FETCH_SIZE WRITE_SIZE"; do
    rm -rf $R/gpurun_out/pmc_gl2
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_gl2 -o p -- python3 /tmp/gram_one.py > /dev/null 2>&1 || { echo "$mode $ctr: profiler failed"; continue; }
    python3 - "$mode" <<'PY'
import csv, collections, sys, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gl2"
fs = [os.path.join(dp, x) for dp, dn, fn in os.walk(root) for x in fn if x.endswith("counter_collection.csv")]
if not fs:
    print(sys.argv[1], "no counter file"); sys.exit(0)
agg = collections.defaultdict(float); cnt = collections.defaultdict(set); dims = collections.defaultdict(lambda: collections.defaultdict(float))
cols = None
for r in csv.DictReader(open(fs[0])):
    if cols is None: cols = list(r.keys())
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not ("gram_kernel" in k or "dgemm2_kernel<0, 0>" in k): continue
    key = (k[:28], r["Counter_Name"])
    agg[key] += float(r["Counter_Value"]); cnt[key].add(r["Dispatch_Id"])
    for c in r:
        if c.upper().startswith("DIMENSION") or c in ("XCC", "Xcc", "Agent_Id"):
            dims[key][c + "=" + r[c]] += float(r[c] and r["Counter_Value"] or 0)
for key in sorted(agg):
    n = max(1, len(cnt[key]))
    print("%-5s %-28s %-20s %2d calls  %.4g per call" % (sys.argv[1], key[0], key[1], n, agg[key] / n))
if cols: print("      columns:", ",".join(cols)[:300])
PY
  done
done
rm -rf $R/gpurun_out/pmc_gl2
