#!/usr/bin/env python3
"""gemm2_abl.py - developer tool (see gemm2_abl.sh): time of the n^3 products of the C2 assembly with the loaded engine variant."""
import ctypes as C
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hb)
lib = hb.ulib()
B_LOWTRI, A_LOWTRI, REMAP, LOWER = 4, 2, 16, 1


def run(name, M, N, K, batch, flags, reps, layB=1, splitk=1):
    used, nd = C.c_int(0), C.c_longlong(0)
    t0, t1 = C.c_double(0), C.c_double(0)
    rc = lib.hipsdp_dgemm_selfcheck2(0, M, N, K, layB, batch, splitk, flags, C.c_double(1.0), C.c_double(0.0), reps, C.byref(used), C.byref(nd),
                                     C.byref(t0), C.byref(t1))
    print("%-28s rc=%d used=%d ndiff=%d  tile %.3f ms  persistent %.3f ms" % (name, rc, used.value, nd.value, t0.value, t1.value), flush=True)


run("A_stack R (B_LOWTRI)", 500500, 500, 500, 1, B_LOWTRI, 5, 1)
run("Gram W W^T (LOWER, split-K)", 1001, 1001, 250000, 1, LOWER, 3, 0, 16)
