"""developer script: the block-Jacobi decomposition (n > 128) on low-rank and full-rank matrices - time per call and residuals.  Device
timings first, host reference decompositions afterwards (idle BLAS threads keep spinning and slow the launching thread down).
HIPSDP_BJ_SORT=0: without the sorting of the coordinates by their diagonal entries."""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
rng = np.random.default_rng(3)
res = []
for n, r in ((200, 20), (200, 200), (300, 10), (500, 50), (500, 500), (257, 1), (130, 129), (400, 399)):
    B = rng.standard_normal((n, r)); W = B @ B.T - 0.01 * np.eye(n)
    hb.syev(W)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); lam, V = hb.syev(W); ts.append(time.perf_counter() - t0)
    res.append((n, r, W, lam, V, min(ts)))
for n, r, W, lam, V, dt in res:
    t0 = time.perf_counter(); ev = np.linalg.eigvalsh(W); th = time.perf_counter() - t0
    print("n %3d rank %3d: %7.2f ms (host eigvalsh %6.1f ms)  |lam - ref| %.1e  |V V^T - I| %.1e  |V W V^T - L| %.1e" % (
        n, r, 1e3 * dt, 1e3 * th, np.abs(lam - ev).max() / np.abs(ev).max(), np.abs(V @ V.T - np.eye(n)).max(),
        np.abs(V @ W @ V.T - np.diag(lam)).max() / np.abs(ev).max()))
