#!/usr/bin/env python3
"""gemm4_lab.py - developer tool: the strip kernel of the two triangular Schur products (csrc/dgemm4.hip) against the tile kernel:
bitwise comparison and time per product (hipsdp_dgemm_selfcheck2).  usage: gemm4_lab.py [small|c2|t1] ..."""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hb)
lib = hb.ulib()
B_LOWTRI, A_LOWTRI, REMAP = 4, 2, 16


def run(name, M, N, K, batch, flags, reps):
    used, nd = C.c_int(0), C.c_longlong(0)
    t0, t1 = C.c_double(0), C.c_double(0)
    rc = lib.hipsdp_dgemm_selfcheck2(0, M, N, K, 1, batch, 1, flags, C.c_double(1.0), C.c_double(0.0), reps, C.byref(used), C.byref(nd),
                                     C.byref(t0), C.byref(t1))
    full = 2.0 * M * N * K * batch
    msg = "%-28s M=%7d N=%5d K=%5d batch=%5d flags=%2d rc=%d used=%d ndiff=%d" % (name, M, N, K, batch, flags, rc, used.value, nd.value)
    if reps > 0 and rc == 0:
        msg += "  tile %.3f ms  default %.3f ms  (%.1f -> %.1f TFLOP/s of the full product)" % (
            t0.value, t1.value, full / t0.value * 1e-9, full / t1.value * 1e-9)
    print(msg, flush=True)
    return rc == 0 and nd.value == 0


A_UPTRI = 256


def run(name, M, N, K, batch, flags, reps, layB=1):
    used, nd = C.c_int(0), C.c_longlong(0)
    t0, t1 = C.c_double(0), C.c_double(0)
    rc = lib.hipsdp_dgemm_selfcheck2(0, M, N, K, layB, batch, 1, flags, C.c_double(1.0), C.c_double(0.0), reps, C.byref(used), C.byref(nd),
                                     C.byref(t0), C.byref(t1))
    full = 2.0 * M * N * K * batch
    msg = "%-30s M=%5d N=%5d K=%5d batch=%5d flags=%3d layB=%d rc=%d used=%d ndiff=%d" % (name, M, N, K, batch, flags, layB, rc, used.value, nd.value)
    if reps > 0 and rc == 0:
        msg += "  tile %.3f ms  default %.3f ms  (%.1f -> %.1f TFLOP/s of the full product)" % (
            t0.value, t1.value, full / t0.value * 1e-9, full / t1.value * 1e-9)
    print(msg, flush=True)
    return rc == 0 and nd.value == 0


what = sys.argv[1:] or ["small"]
ok = True
if "small" in what:
    for n, b in ((500, 101), (1000, 40), (362, 160), (70, 600), (514, 70), (128, 200), (1026, 24)):
        ok &= run("G T_j,   T_j rows of K", n, n, n, b, A_LOWTRI | REMAP, 0, 1)
        ok &= run("G T_j,   T_j K contiguous", n, n, n, b, A_LOWTRI | REMAP, 0, 0)
        ok &= run("R^T A_j, A_j rows of K", n, n, n, b, A_UPTRI | REMAP, 0, 1)
        ok &= run("R^T A_j, A_j K contiguous", n, n, n, b, A_UPTRI | REMAP, 0, 0)
if "c2" in what:
    ok &= run("C2 first product  R^T A_j", 500, 500, 500, 1001, A_UPTRI | REMAP, 5, 1)
    ok &= run("C2 second product G T_j", 500, 500, 500, 1001, A_LOWTRI | REMAP, 5, 0)
    ok &= run("C2 old first product (stack)", 500500, 500, 500, 1, B_LOWTRI, 5, 1)
    ok &= run("C2 old second product", 500, 500, 500, 1001, A_LOWTRI | REMAP, 5, 1)
if "t1" in what:
    ok &= run("T1 first product  R^T A_j", 1000, 1000, 1000, 2001, A_UPTRI | REMAP, 3, 1)
    ok &= run("T1 second product G T_j", 1000, 1000, 1000, 2001, A_LOWTRI | REMAP, 3, 0)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
