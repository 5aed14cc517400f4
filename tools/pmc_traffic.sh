# HBM traffic of one bench solve: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > /dev/null 2>$R/gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > /dev/null 2>$R/gpurun_out/pmc_write.err
ls -la $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
