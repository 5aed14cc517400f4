# developer tool: ms per IPM iteration of mid-size dense blocks (between the B&B-sized and the bench-sized regime), repeated runs
run() { timeout -k 10 120 python bench.py --n $1 --m $2 --steps ${4:-5} --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$3', d['config']['n'], d['config']['m'], 'ms/iter %.3f' % (d['ms_per_step']/d['iterations_per_solve']), 'schur ms %.3f' % d['roofline']['avg_assembly_ms'])"; }
for i in 1 2 3 4; do run 200 300 rep; done
for i in 1 2 3; do run 200 300 rep20 20; done
for i in 1 2 3; do HIPSDP_GEMM_V1=1 run 200 300 gemmv1-20 20; done
for i in 1 2 3; do HIPSDP_ONEQUEUE=1 run 200 300 onequeue-20 20; done
