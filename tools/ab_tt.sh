for i in 1 2; do
python -m pytest tests/test_gpu_bnb.py -m gpu -q -s -k "reproduces_short_solu and TT" 2>&1 | grep -E "optimum" | sed 's/^/NEW /'
cp scip-sdp_amd/lib/libhipsdp.so /tmp/keep.so; cp scip-sdp_amd/lib/libhipsdp_old.so scip-sdp_amd/lib/libhipsdp.so
python -m pytest tests/test_gpu_bnb.py -m gpu -q -s -k "reproduces_short_solu and TT" 2>&1 | grep -E "optimum" | sed 's/^/OLD /'
cp /tmp/keep.so scip-sdp_amd/lib/libhipsdp.so
done
