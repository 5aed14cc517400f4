#!/bin/bash
# developer tool: one mid-size shape under several environment settings, repeated (ms per iteration, Schur share)
cd $GRAFT_REPO_ROOT
n=${1:-300}; m=${2:-200}; shift 2
[ $# -eq 0 ] && set -- "HIPSDP_NONE=0"
for v in "$@"; do
for rep in 1 2 3; do
env $v timeout -k 10 200 python3 bench.py --n $n --m $m --steps 3 --warmup 1 --no-cpu --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v n %d m %d ms/iter %.3f schur %.3f iters %.1f' % (d['config']['n'], d['config']['m'], d['ms_per_step']/d['iterations_per_solve'], d['roofline']['avg_assembly_ms'], d['iterations_per_solve']))"
done; done
