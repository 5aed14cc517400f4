cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $R/gpurun_out/pmc_$1 -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>$R/gpurun_out/pmc_$1.err; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS"
run b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS"
run c "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCC_HIT TCC_MISS TA_BUSY"
ls -la $R/gpurun_out/pmc_*/ | head -30
