"""developer tool: where the wavefronts of the paired-band kernel spend their cycles (library built with make EXTRA=-DG5_PROF).
usage: python3 tests/devtools/tri5_prof.py [which]   which = 0: A_stack R at n = 500, 1: G T_j at n = 500, 2 / 3: the same at n = 1000"""
import os, sys, importlib.util, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
A_LOW, B_LOW, REMAP = 2, 4, 16
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
case = [(500500, 500, 500, 1, 1, B_LOW), (500, 500, 500, 1, 1001, A_LOW | REMAP),
        (2001000, 1000, 1000, 1, 1, B_LOW), (1000, 1000, 1000, 1, 2001, A_LOW | REMAP)][which]
M, N, K, layB, batch, flags = case
used, nd, md, nr, t0, t1 = hb.dgemm_selfcheck3(M, N, K, layB=layB, batch=batch, flags=flags, reps=3)
print("case", case, "ms tile %.3f fast %.3f maxdiff %.2e" % (t0, t1, md))
buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
rc = hb.ulib().hs_dgemm5_prof_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)))
assert rc == 0
P = buf.reshape(512, 4, 8).astype(np.float64)
tot = P[:, :, 7]
names = ["full: DMA wait", "full: barrier", "band: DMA wait", "band: barrier", "full stages", "band double stages", "epilogue", "total"]
print("per wavefront, mean over %d wavefronts (cycles of s_memtime; share of total)" % (512 * 4))
for i, nm in enumerate(names):
    v = P[:, :, i]
    if i in (4, 5):
        print("  %-20s %12.0f" % (nm, v.mean()))
    else:
        print("  %-20s %12.0f  %5.1f %%   (min %.0f max %.0f)" % (nm, v.mean(), 100 * v.mean() / tot.mean(), v.min(), v.max()))
for w in range(4):
    print("  wave %d: full wait %.0f barrier %.0f | band wait %.0f barrier %.0f | epilogue %.0f | total %.0f" %
          (w, P[:, w, 0].mean(), P[:, w, 1].mean(), P[:, w, 2].mean(), P[:, w, 3].mean(), P[:, w, 6].mean(), P[:, w, 7].mean()))
nf, nb = P[:, :, 4].mean(), P[:, :, 5].mean()
print("  per full stage: wait %.0f barrier %.0f ; per double stage: wait %.0f barrier %.0f ; cycles per stage-equivalent %.0f" %
      (P[:, :, 0].mean() / max(nf, 1), P[:, :, 1].mean() / max(nf, 1), P[:, :, 2].mean() / max(nb, 1), P[:, :, 3].mean() / max(nb, 1),
       (tot.mean() - P[:, :, 6].mean()) / (nf + 1.125 * nb)))
