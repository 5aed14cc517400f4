#!/bin/bash
# developer tool: ms per IPM iteration over a grid of (n, m) between the B&B-sized and the bench-sized regime, with the share of the
# Schur assembly and the rate the iteration would have if it ran at the matrix peak (to spot sizes where a path falls off a cliff)
cd $GRAFT_REPO_ROOT
for n in ${NS:-32 48 64 65 96 128 200 300 400}; do
for m in ${MS:-50 100 200 500 1000}; do
timeout -k 10 200 python3 bench.py --n $n --m $m --steps ${STEPS:-3} --warmup ${WARM:-1} --no-cpu --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
n=d['config']['n']; m=d['config']['m']; it=d['iterations_per_solve']
msit=d['ms_per_step']/it
fl=4.0*(m+1)*n**3+(m+1)**2*n**2+(m**3)/3.0
print('n %4d m %5d  iters %4.1f  ms/iter %8.3f  schur ms %8.3f  ideal(78.6TF) ms %7.3f  ok %s' % (n, m, it, msit, d['roofline']['avg_assembly_ms'], fl/78.6e9, d['solution_check']['status_optimal_and_objective_matches_planted_optimum']))" || echo "n $n m $m FAILED"
done; done
