cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mid -o p -- python3 $GRAFT_REPO_ROOT/bench.py --n ${1:-300} --m ${2:-200} --steps 3 --warmup 1 --no-cpu --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_mid.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_mid/p_kernel_stats.csv gpurun_out/bnb_kernel_stats.csv
python3 tools/show_kernel_stats.py | head -30
python3 -c "
import json; d=json.load(open('gpurun_out/prof_mid.json')); print('ms/iter', d['ms_per_step']/d['iterations_per_solve'], 'iters', d['iterations_per_solve'], 'schur', d['roofline']['avg_assembly_ms'])"
rm -rf gpurun_out/prof_mid
