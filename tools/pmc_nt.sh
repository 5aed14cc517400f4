#!/bin/bash
# pmc_nt.sh - developer tool: L2-miss traffic (FETCH_SIZE, WRITE_SIZE) and time of the paired-band products at the bench shapes with the
# natural list order / plain stores (HIPSDP_GEMM_ORDER=0) and the default (sets of panels, entries in pairs, non-temporal stores)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "0" "1"; do
  export HIPSDP_GEMM_ORDER=$cfg
  echo "== HIPSDP_GEMM_ORDER=$cfg: times"
  timeout -k 10 120 python3 $R/tests/devtools/tri5_check.py 5 2>&1 | cut -c1-60,100-
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_nt
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_nt -o p -- python3 $R/tests/devtools/tri5_check.py 2 > /dev/null 2>&1
    python3 - $c <<'PY'
import csv, collections, sys, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_nt"
f = [os.path.join(dp, x) for dp, dn, fn in os.walk(root) for x in fn if x.endswith("counter_collection.csv")][0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != sys.argv[1]: continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "dgemm5" in k:
        agg[k].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    mul = 2 if sys.argv[1] == "FETCH_SIZE" else 1
    print("  %-12s %-28s %2d calls, GB per call:" % (sys.argv[1], k[:28], len(v)), " ".join("%.2f" % (mul * x * 1024 / 1e9) for x in v))
PY
  done
done
rm -rf $R/gpurun_out/pmc_nt
