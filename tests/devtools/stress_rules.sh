#!/bin/bash
# developer tool: the 400-problem stress family (with the settings ladder) under pivot rules 2 and 3, engine against oracle
cd $GRAFT_REPO_ROOT
for rule in 2 3; do
  echo "== HIPSDP_PIVOT_RULE=$rule"
  HIPSDP_PIVOT_RULE=$rule STRESS_LADDER=1 timeout -k 10 500 python3 tests/devtools/stress_gpu.py 400 0 2>&1 | tail -12
done
