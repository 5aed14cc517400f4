#!/bin/bash
# gram_plan_sweep.sh - developer tool: bench line at C2 under several K-slice plans of the Gram kernel (HIPSDP_GRAM_PLAN="so sd")
for rep in 1 2; do
for plan in "" "16 8" "14 14" "15 12" "16 10" "17 4"; do
  if [ -z "$plan" ]; then unset HIPSDP_GRAM_PLAN; else export HIPSDP_GRAM_PLAN="$plan"; fi
  python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('plan [$plan]', 'solves/s', round(d['value'],3), 'assembly ms', round(d['roofline']['avg_assembly_ms'],3))
"
done; done
