# developer tool: which class of products makes a stress seed differ (HIPSDP_GEMM_V1_MASK routes classes to the tile kernel)
export STRESS_BIG=1
for seed in ${SEEDS:-1000 1010}; do
for mask in 0 1 2 4 8 16 32; do
echo "== seed $seed mask $mask: $(HIPSDP_GEMM_V1_MASK=$mask python tests/devtools/stress_gpu.py 1 $seed 2>&1 | tail -1)"
done; done
