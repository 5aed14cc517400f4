"""cpu_ref_dense.py - TEST / BENCH INFRASTRUCTURE (oracle): ctypes view of oracle/libcpu_ref_dense.so, the C restatement of ipm_ref.hsd_solve for
one dense block on the host's BLAS / LAPACK (oracle/cpu_ref_dense.c), linked against scipy's bundled OpenBLAS (symbols scipy_dgemm_ ...).
Used by bench.py's cpu_baseline leg and by tests/test_cpu_ref.py; never by the product."""
import ctypes as C
import glob
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libcpu_ref_dense.so")


class DenseInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("pobj", C.c_double), ("dobj", C.c_double), ("pinf", C.c_double),
                ("dinf", C.c_double), ("gap", C.c_double), ("mu", C.c_double), ("tau", C.c_double), ("kappa", C.c_double),
                ("schur_seconds", C.c_double), ("total_seconds", C.c_double)]


def openblas():
    """path of scipy's bundled OpenBLAS (None when this image has none)"""
    try:
        import scipy
    except ImportError:
        return None
    g = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    return os.path.abspath(g[0]) if g else None


def build():
    blas = openblas()
    if blas is None:
        raise RuntimeError("no OpenBLAS with scipy_-prefixed symbols in this image")
    src = os.path.join(_HERE, "cpu_ref_dense.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", LIB, src, blas, "-Wl,-rpath," + os.path.dirname(blas), "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def blas_name():
    b = openblas()
    return os.path.basename(b) if b else "none"


def solve(b, A, gaptol=1e-5, feastol=1e-5, pabstol=0.0, infeastol=1e-7, gamma=0.98, maxiter=100, settings=0, threads=0):
    """b[m], A[(m + 1), n, n] (A[0] = constant matrix).  Returns (DenseInfo, y).  threads <= 0: the BLAS's own default."""
    PD = C.POINTER(C.c_double)
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    m, n = b.shape[0], A.shape[1]
    assert A.shape == (m + 1, n, n)
    y = np.zeros(max(1, m))
    info = DenseInfo()
    rc = lib().cpu_ref_dense_solve(C.c_int(m), C.c_int(n), A.ctypes.data_as(PD), b.ctypes.data_as(PD), C.c_double(gaptol), C.c_double(feastol),
                                   C.c_double(pabstol), C.c_double(infeastol), C.c_double(gamma), C.c_int(maxiter), C.c_int(settings),
                                   C.c_int(threads), y.ctypes.data_as(PD), C.byref(info))
    if rc != 0:
        raise MemoryError("cpu_ref_dense_solve: out of memory")
    return info, y[:m]
