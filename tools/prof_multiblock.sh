cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mb -o p -- python3 $GRAFT_REPO_ROOT/tests/devtools/multiblock_time.py ${1:-20,50,500} > $GRAFT_REPO_ROOT/gpurun_out/prof_mb.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_mb/p_kernel_stats.csv gpurun_out/bnb_kernel_stats.csv
python3 tools/show_kernel_stats.py | head -${2:-32}
cat gpurun_out/prof_mb.txt
rm -rf gpurun_out/prof_mb
