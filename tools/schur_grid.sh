# developer tool: Schur assembly time / algorithmic TFLOP/s over a grid of (n, m) - looks for cliffs in the kernel selection
for n in 64 96 128 192 256 384 512; do for m in 100 255 256 300 500 700 1000 2000; do
timeout -k 10 120 python bench.py --n $n --m $m --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n %4d m %5d  schur %8.3f ms  %6.1f TF/s  iter %7.3f ms  schur share %.2f' % (d['config']['n'], d['config']['m'], d['roofline']['avg_assembly_ms'], d['roofline']['achieved'], d['ms_per_step']/d['iterations_per_solve'], d['roofline']['schur_share_of_solve_time']))"
done; done
