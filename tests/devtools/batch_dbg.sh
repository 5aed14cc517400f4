cd $GRAFT_REPO_ROOT
HIPSDP_BATCH_TIMES=1 python3 tests/devtools/bnb_gpu_run.py example_TT.dat-s.gz 2>&1 | tail -16
for sk in 0 1 2 4 8 16 32 64 128; do echo -n "skip $sk: "; HIPSDP_BATCH_SKIP=$sk python3 tests/devtools/bnb_gpu_run.py example_CLS.dat-s.gz 2>&1 | tail -1; done
