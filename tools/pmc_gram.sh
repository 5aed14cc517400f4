#!/bin/bash
# pmc_gram.sh - developer tool: L2-miss traffic (FETCH_SIZE) of the Gram product at the bench shape through the K-sliced tile kernel and
# through the Gram kernel under several plans (HIPSDP_GRAM_PLAN="so sd")
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gram_one.py <<'PY'
import os, sys, importlib.util
ROOT = os.environ["GRAFT_REPO_ROOT"]
spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
print(hb.gram_selfcheck(1001, 250000, reps=3))
PY
for plan in "" "14 14" "15 15" "16 8"; do
  rm -rf $R/gpurun_out/pmc_gram
  if [ -z "$plan" ]; then unset HIPSDP_GRAM_PLAN; else export HIPSDP_GRAM_PLAN="$plan"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_gram -o p -- python3 /tmp/gram_one.py > /dev/null 2>&1
  python3 - "$plan" <<'PY'
import csv, collections, sys, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gram"
f = [os.path.join(dp, x) for dp, dn, fn in os.walk(root) for x in fn if x.endswith("counter_collection.csv")][0]
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in agg:
    if "gram_kernel" in k or "dgemm2_kernel<0, 0>" in k:
        print("plan '%s' %-40s %2d calls  fetch (x2) %.3f GB per call" % (sys.argv[1], k[:40], len(cnt[k]), 2 * agg[k] * 1024 / 1e9 / len(cnt[k])))
PY
done
rm -rf $R/gpurun_out/pmc_gram
