# developer tool: planted instances of unusual aspect ratios through bench.py (exit code 2 / "ok false" = wrong optimum)
for s in "100 3000" "64 2000" "150 5000" "96 4500" "300 100" "700 64" "1000 40" "65 1" "130 2" "257 511" "384 3000"; do set -- $s
timeout -k 10 300 python bench.py --n $1 --m $2 --steps 1 --warmup 0 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['solution_check']; print('n %4d m %5d ok %s iters %.0f ms %.1f obj %.9g planted %.9g' % (d['config']['n'], d['config']['m'], c['status_optimal_and_objective_matches_planted_optimum'], d['iterations_per_solve'], d['ms_per_step'], c['objective'], c['planted_optimum']))" || echo "n $1 m $2 FAILED"
done
