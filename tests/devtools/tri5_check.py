"""developer tool: the paired-band kernel of the triangular products (hs_dgemm5_kernel) against the tile kernel: largest
difference and times.  usage: python3 tests/devtools/tri5_check.py [reps]"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
A_LOW, B_LOW, REMAP = 2, 4, 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = [
    (4000, 500, 500, 1, 1, B_LOW),
    (50000, 500, 500, 1, 1, B_LOW),
    (50000, 372, 500, 1, 1, B_LOW),
    (40000, 256, 384, 1, 1, B_LOW),
    (500, 500, 500, 1, 40, A_LOW | REMAP),
    (500, 500, 500, 0, 40, A_LOW | REMAP),
    (384, 244, 384, 1, 60, A_LOW | REMAP),
    (1000, 1000, 1000, 1, 16, A_LOW | REMAP),
    (30000, 1000, 1000, 1, 1, B_LOW),
    (30000, 1000, 1000, 0, 1, B_LOW),
]
if reps > 0:
    cases = [(500500, 500, 500, 1, 1, B_LOW), (500, 500, 500, 1, 1001, A_LOW | REMAP),
             (2001000, 1000, 1000, 1, 1, B_LOW), (1000, 1000, 1000, 1, 2001, A_LOW | REMAP)]
for (M, N, K, layB, batch, flags) in cases:
    used, nd, md, nr, t0, t1 = hb.dgemm_selfcheck3(M, N, K, layB=layB, batch=batch, flags=flags, reps=reps)
    print("M %7d N %5d K %5d layB %d batch %4d flags %2d: used %d, differing %d, max |diff| %.3e, not reproduced %d, ms tile %.3f fast %.3f" %
          (M, N, K, layB, batch, flags, used, nd, md, nr, t0, t1), flush=True)
