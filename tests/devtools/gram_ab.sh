#!/bin/bash
# developer tool: bitwise GEMM tests, then the solve at the bench shape (assembly time, solves/s) and the per-kernel times,
# for each of the environment settings given as arguments (default: one run with the defaults)
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_gpu_units.py -x -q -m gpu -k "gemm or schur" > gpurun_out/gram_ab_tests.log 2>&1 || { tail -30 gpurun_out/gram_ab_tests.log; exit 1; }
tail -2 gpurun_out/gram_ab_tests.log
[ $# -eq 0 ] && set -- "HIPSDP_NONE=0"
k=0
for v in "$@"; do
  k=$((k+1))
  echo "== $v"
  export $v
  timeout -k 10 200 python3 bench.py --no-cpu --no-extras --steps 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('solves/s %.3f  ms/step %.2f  assembly ms %.3f  frac %.3f iters %.1f ok %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_assembly_ms'], d['roofline']['frac'], d['iterations_per_solve'], d['solution_check']['status_optimal_and_objective_matches_planted_optimum']))"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gram_ab_prof$k -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find gpurun_out/gram_ab_prof$k -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us")
PY
  rm -rf gpurun_out/gram_ab_prof$k
  unset ${v%%=*}
done
