# developer tool: the sharded solve at bench size (n=500, m=1000) with 2 processes on one device through the host-staged transport,
# replicated matrices (column slices + all-reduce, sharded passes) and matrices sharded by variable; compares with one process
cd $GRAFT_REPO_ROOT
T=/tmp/mbs; mkdir -p $T
export HIPSDP_TEST_STAGING=$((64<<20))
python tests/multi_worker.py 0 1 /mbs1 500 1000 0 $T/one.json gen || exit 1
for mode in gen vars-gen; do
  python tests/multi_worker.py 0 2 /mbs_$mode 500 1000 0 $T/${mode}_0.json $mode & p0=$!
  python tests/multi_worker.py 1 2 /mbs_$mode 500 1000 0 $T/${mode}_1.json $mode & p1=$!
  wait $p0 || exit 1; wait $p1 || exit 1
done
python - <<'PY'
import json, numpy as np
one = json.load(open("/tmp/mbs/one.json"))
for mode in ("gen", "vars-gen"):
    for r in (0, 1):
        d = json.load(open("/tmp/mbs/%s_%d.json" % (mode, r)))
        print(mode, "rank", r, "status", d["status"], "iterations", d["iterations"], "(one process: %d)" % one["iterations"],
              "max |y - y1| %.2e" % np.max(np.abs(np.array(d["y"]) - np.array(one["y"]))), "dobj %.12g vs %.12g" % (d["dobj"], one["dobj"]))
PY
