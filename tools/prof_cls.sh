cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cls -o cls -- python3 $GRAFT_REPO_ROOT/tests/devtools/solve1_tt.py example_CLS.dat-s.gz > $GRAFT_REPO_ROOT/gpurun_out/cls.txt 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_cls/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_cls/*/*kernel_stats.csv 2>/dev/null | head -1)
head -30 $f | cut -c1-150
tail -2 $GRAFT_REPO_ROOT/gpurun_out/cls.txt
