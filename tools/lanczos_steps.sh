# developer tool: solves/s and iterations of the bench instance against the number of Lanczos steps per step-length estimate
run() { timeout -k 10 200 python bench.py --n $1 --m $2 --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanczos $3: n %d m %d  %.3f solves/s  iters %.1f  ms/iter %.3f' % (d['config']['n'], d['config']['m'], d['value'], d['iterations_per_solve'], d['ms_per_step']/d['iterations_per_solve']))"; }
for L in 24 16 12 8; do HIPSDP_LANCZOS=$L run 500 1000 $L; done
for L in 24 16 12 8; do HIPSDP_LANCZOS=$L run 200 300 $L; done
for L in 24 12; do HIPSDP_LANCZOS=$L run 1000 2000 $L; done
