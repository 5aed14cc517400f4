#!/usr/bin/env python3
"""lapack_small_time.py - developer tool: wall time per call of the SCIPlapack* entry points at the sizes their callers use
(cons_sdp.c: smallest eigenvalue of blocks of 2-50 rows, dozens of calls per node), transfers included; numpy / LAPACK on the
host's cores beside it."""
import ctypes as C, importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
lib = hb.lib()
PD = C.POINTER(C.c_double)
pd = lambda a: a.ctypes.data_as(PD)
print("%5s %20s %20s %20s %16s %14s %14s %14s %14s" % ("n", "IthEigenvalue(1) us", "IthEigenvalue+vec us", "EigenvectorDecomp us", "numpy eigvalsh us",
      "MatMatMult us", "numpy A@B us", "MatVecMult us", "numpy A@x us"))
for n in (2, 5, 10, 16, 20, 30, 43, 50, 64, 65, 100, 128):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n)); A = (G + G.T)
    a = A.reshape(-1).copy()
    val = C.c_double(0.0); vec = np.zeros(n); lam = np.zeros(n); V = np.zeros(n * n)
    reps = 300 if n <= 64 else 30
    def timeit(f):
        for _ in range(3): f()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        return 1e6 * (time.perf_counter() - t0) / reps
    t1 = timeit(lambda: lib.SCIPlapackComputeIthEigenvalue(None, 0, n, pd(a), 1, C.byref(val), None))
    t2 = timeit(lambda: lib.SCIPlapackComputeIthEigenvalue(None, 1, n, pd(a), 1, C.byref(val), pd(vec)))
    t3 = timeit(lambda: lib.SCIPlapackComputeEigenvectorDecomposition(None, n, pd(a.copy()), pd(lam), pd(V)))
    t4 = timeit(lambda: np.linalg.eigvalsh(A))
    Bm = rng.standard_normal((n, n)); bflat = Bm.reshape(-1).copy(); cres = np.zeros(n * n); xv = rng.standard_normal(n); yv = np.zeros(n)
    t5 = timeit(lambda: lib.SCIPlapackMatrixMatrixMult(n, n, pd(a), 0, n, n, pd(bflat), 0, pd(cres)))
    t6 = timeit(lambda: A @ Bm)
    t7 = timeit(lambda: lib.SCIPlapackMatrixVectorMult(n, n, pd(a), pd(xv), pd(yv)))
    t8 = timeit(lambda: A @ xv)
    print("%5d %20.1f %20.1f %20.1f %16.1f %14.1f %14.1f %14.1f %14.1f" % (n, t1, t2, t3, t4, t5, t6, t7, t8))
