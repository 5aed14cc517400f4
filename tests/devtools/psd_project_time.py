"""developer script: wall time of the fused PSD projection (hipsdp_psd_project, literal mode) per call.
usage: python tests/devtools/psd_project_time.py"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
rng = np.random.default_rng(3)
for n, r in ((10, 2), (20, 3), (43, 5), (64, 8), (128, 16), (200, 20)):
    B = rng.standard_normal((n, r)); W = B @ B.T - 0.01 * np.eye(n)
    iu = np.triu_indices(n)
    row, col, val = iu[0].astype(np.int32), iu[1].astype(np.int32), W[iu]
    hb.psd_project(n, row, col, val, 1e-4, 1e-9, 0)
    t0 = time.perf_counter()
    for _ in range(20):
        hb.psd_project(n, row, col, val, 1e-4, 1e-9, 0)
    dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        lam, V = np.linalg.eigh(W); (V * np.maximum(lam, 1e-4)) @ V.T
    dh = (time.perf_counter() - t0) / 20
    print("n %3d rank %2d: %8.1f us per projection on the device, %8.1f us with numpy on the host" % (n, r, dt * 1e6, dh * 1e6))
