"""developer script: wall time of the full decomposition (hipsdp_syev) at 64 ... 128 rows, one launch against the block Jacobi
(HIPSDP_SYEV_JACOBI is read once per process: run twice to compare).  usage: python tests/devtools/syev_mid_time.py"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
rng = np.random.default_rng(1)
sizes = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (32, 64, 65, 80, 96, 100, 112, 128)
mats = {}
for n in sizes:
    G = rng.standard_normal((n, n)); mats[n] = G + G.T
dev = {}
for n in sizes:                     # (all device timings first: the host library's worker threads keep spinning after a call)
    W = mats[n]
    hb.syev(W)
    t0 = time.perf_counter()
    for _ in range(20):
        lam, V = hb.syev(W)
    dev[n] = ((time.perf_counter() - t0) / 20, np.abs(V @ W @ V.T - np.diag(lam)).max() / np.abs(lam).max(), np.abs(V @ V.T - np.eye(n)).max())
for n in sizes:
    W = mats[n]
    t0 = time.perf_counter()
    for _ in range(20):
        np.linalg.eigh(W)
    dh = (time.perf_counter() - t0) / 20
    print("n %3d: %8.1f us per call (host eigh %7.1f us), residual %.1e, orthogonality %.1e" % (n, dev[n][0] * 1e6, dh * 1e6, dev[n][1], dev[n][2]))
