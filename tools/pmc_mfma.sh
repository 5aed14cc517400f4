# FP64-MFMA utilisation of the Schur kernels: SQ counters in one rocprofv3 pass, GRBM in another (kernel trace only beside --pmc)
# usage (on the GPU box): bash tools/pmc_mfma.sh TAG [bench args]; results under gpurun_out/pmc_mfma_TAG/
tag=${1:-c2}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mfma_$tag
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras "$@" > $O/bench_sq.json 2> $O/sq.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras "$@" > $O/bench_grbm.json 2> $O/grbm.err
python3 $R/tools/pmc_mfma_table.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
find $O -name "*.csv" -size +8M -delete
