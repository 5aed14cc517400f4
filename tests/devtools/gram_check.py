"""developer tool: the Gram kernel (csrc/gram.hip) against the K-sliced tile kernels: difference, reproducibility, times.
usage: python3 tests/devtools/gram_check.py [reps]"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = [(1001, 30000), (401, 25000), (700, 20000), (1001, 31000), (300, 40000), (2001, 20000), (1024, 16384)]
if reps > 0:
    cases = [(1001, 250000), (2001, 1000000)]
import ctypes as C
for (M, K) in cases:
    no, nd, ni, sp = C.c_int(0), C.c_int(0), C.c_int(0), C.c_double(0.0)
    if hb.ulib().hipsdp_gram_plan_info(0, M, C.c_longlong(K), 64, C.byref(no), C.byref(nd), C.byref(ni), C.byref(sp)) == 0:
        print("plan: %d / %d partial tiles per off-diagonal / diagonal tile, %d items, estimated span %.4f" % (no.value, nd.value, ni.value, sp.value), flush=True)
    used, md, nr, t0, t1 = hb.gram_selfcheck(M, K, reps=reps)
    print("M %5d K %8d: used %d, max |diff| %.3e, not reproduced %d, ms tile path %.3f Gram kernel %.3f" % (M, K, used, md, nr, t0, t1), flush=True)
