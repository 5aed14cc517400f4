#!/bin/bash
# developer tool: A/B of the Schur assembly variants inside the solve (same box, same run)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "HIPSDP_GEMM4=0 HIPSDP_SCHUR_STACK=1" "HIPSDP_SCHUR_STACK=1" "HIPSDP_GEMM4=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 bench.py --no-cpu --no-extras --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('solves/s %.3f  ms/step %.2f  assembly ms %.3f  iters %.1f ok %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_assembly_ms'], d['iterations_per_solve'], d['solution_check']['status_optimal_and_objective_matches_planted_optimum']))"
done
done
