"""cpu_ref.py - TEST / BENCH INFRASTRUCTURE (oracle): ctypes view of oracle/libcpu_ref.so, the plain-C restatement of ipm_ref.hsd_solve
for B&B-sized problems (oracle/cpu_ref.c).  Used by bench.py's CPU leg and by tests/test_cpu_ref.py; never by the product."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libcpu_ref.so")


class CpuInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("pobj", C.c_double), ("dobj", C.c_double), ("pinf", C.c_double),
                ("dinf", C.c_double), ("gap", C.c_double), ("mu", C.c_double), ("tau", C.c_double), ("kappa", C.c_double)]


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "cpu_ref.c")):
        subprocess.check_call(["gcc", "-O3", "-shared", "-fPIC", "-o", LIB, os.path.join(_HERE, "cpu_ref.c"), "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def solve(prob, gaptol=1e-5, feastol=1e-5, pabstol=0.0, infeastol=1e-7, gamma=0.98, maxiter=100, settings=0):
    """prob: ipm_ref.CoreProblem.  Returns (CpuInfo, y)."""
    PD = C.POINTER(C.c_double)
    blocks = [np.ascontiguousarray(A, dtype=np.float64) for A in prob.blocks]
    ns = (C.c_int * max(1, len(blocks)))(*[A.shape[1] for A in blocks])
    ptrs = (PD * max(1, len(blocks)))(*[A.ctypes.data_as(PD) for A in blocks])
    Dext = np.ascontiguousarray(np.concatenate([prob.c.reshape(-1, 1), prob.D], axis=1), dtype=np.float64)
    b = np.ascontiguousarray(prob.b, dtype=np.float64)
    y = np.zeros(max(1, prob.m))
    info = CpuInfo()
    lib().cpu_ref_solve(C.c_int(prob.m), C.c_int(len(blocks)), ns, ptrs, C.c_int(prob.q), Dext.ctypes.data_as(PD), b.ctypes.data_as(PD),
                        C.c_double(gaptol), C.c_double(feastol), C.c_double(pabstol), C.c_double(infeastol), C.c_double(gamma),
                        C.c_int(maxiter), C.c_int(settings), y.ctypes.data_as(PD), C.byref(info))
    return info, y[:prob.m]
