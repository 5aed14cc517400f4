#!/bin/bash
# developer tool: the held batch regions of the B&B-sized regime on and off - same optimum, node count and IPM iteration total
# (the recorded operations do the arithmetic of their launches), and the time of the tree
cd $GRAFT_REPO_ROOT
for inst in example_TT.dat-s.gz example_CLS.dat-s.gz example_small.dat-s example_tightenmatrices.dat-s; do
  f=$(ls tests/golden/instances/ | grep "^${inst%%.*}\." | head -1)
  for b in 0 1 0 1; do
    echo -n "HIPSDP_BATCH=$b: "
    HIPSDP_BATCH=$b timeout -k 10 300 python3 tests/devtools/bnb_gpu_run.py $f 2>&1 | tail -1
  done
done
