#!/bin/bash
# ab_gemm_order.sh - developer tool: bench line at C2 and T1 with the list order / store policy switches of the paired-band kernel
for i in 1 2; do
for cfg in 0 1; do
for size in "500 1000 12" "1000 2000 2"; do
set -- $size
HIPSDP_GEMM_ORDER=$cfg python3 bench.py --n $1 --m $2 --steps $3 --warmup 2 --no-extras --no-cpu 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('order $cfg n $1', 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],2))
"
done; done; done
