#!/usr/bin/env python3
"""syev_time.py [n ...] - developer tool: time and accuracy of the device eigen-decomposition behind SCIPlapackCompute* (hipsdp_syev,
includes the transfers of the n x n matrix and of the eigenvectors) against numpy's LAPACK on the host.  All device timings are
taken before the first host decomposition (idle BLAS threads keep spinning for a while and slow the launching thread down)."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
rng = np.random.default_rng(3)
hb.syev(np.eye(4))
sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 500, 1000]
res = []
for n in sizes:
    G = rng.standard_normal((n, n)); W = G + G.T
    hb.syev(W)
    reps = 20 if n <= 128 else 3
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); lam, V = hb.syev(W); ts.append(time.perf_counter() - t0)
    res.append((n, W, lam, V, min(ts), sorted(ts)[len(ts) // 2]))
for n, W, lam, V, tmin, tmed in res:
    t1 = time.perf_counter(); ev, Q = np.linalg.eigh(W); t2 = time.perf_counter()
    print("n=%5d  device %8.2f ms (median %8.2f)   host eigh %8.1f ms   |lam - ref| %.1e   |V V^T - I| %.1e   |V W V^T - L| %.1e" % (
        n, 1e3 * tmin, 1e3 * tmed, 1e3 * (t2 - t1), np.abs(lam - ev).max() / np.abs(ev).max(), np.abs(V @ V.T - np.eye(n)).max(),
        np.abs(V @ W @ V.T - np.diag(lam)).max() / np.abs(ev).max()), flush=True)
