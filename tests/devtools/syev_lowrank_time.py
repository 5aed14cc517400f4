"""developer script: full decomposition of low-rank matrices (a large cluster of zero eigenvalues: what the PSD projection of the
warm-start producer meets), wall time per call.  usage: python tests/devtools/syev_lowrank_time.py   (HIPSDP_SYEV_MID_FROM=65 for
the round-3 kernel below 65 rows, HIPSDP_SYEV_JACOBI=1 for the block Jacobi)"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
rng = np.random.default_rng(7)
for n, r in ((20, 3), (43, 5), (64, 8), (100, 10), (128, 32), (128, 4)):
    B = rng.standard_normal((n, r)); W = B @ B.T
    hb.syev(W)
    t0 = time.perf_counter()
    for _ in range(20):
        lam, V = hb.syev(W)
    dt = (time.perf_counter() - t0) / 20
    res = np.abs(V @ W @ V.T - np.diag(lam)).max() / np.abs(lam).max()
    print("n %3d rank %2d: %8.1f us per call, residual %.1e, orthogonality %.1e" % (n, r, dt * 1e6, res, np.abs(V @ V.T - np.eye(n)).max()))
