#!/usr/bin/env python3
"""gemm2_scan.py - developer tool: the persistent LDS-DMA GEMM (dgemm2.hip) against the tile GEMM (dgemm.hip) on the shapes the Schur
assembly and the chain products produce for mid-size blocks; prints every shape whose results differ."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
LOWER, A_LOWTRI, B_LOWTRI, XCD, REMAP, NOFAST = 1, 2, 4, 8, 16, 64          # hs_common.h
bad = 0
def run(tag, M, N, K, layB, batch, splitk, flags, beta):
    global bad
    used, nd = hb.dgemm_selfcheck(M, N, K, layB=layB, batch=batch, splitk=splitk, flags=flags, beta=beta)
    if nd != 0:
        bad += 1
        print("DIFF %-10s M %d N %d K %d layB %d batch %d splitk %d flags %d beta %g: v2 used %d, %d entries differ" % (tag, M, N, K, layB, batch, splitk, flags, beta, used, nd), flush=True)
for n in (100, 129, 160, 200, 257, 300):
    for m1 in (130, 201, 258, 301, 401):
        run("stack", m1 * n, n, n, 1, 1, 1, B_LOWTRI, 0.0)
        run("batched", n, n, n, 1, m1, 1, A_LOWTRI | REMAP, 0.0)
        run("stackU", m1 * n, n, n, 1, 1, 1, REMAP, 0.0)
        run("batchedU", n, n, n, 1, m1, 1, REMAP, 0.0)
        for sk in (1, 2, 7, 16, 25, 39, 48, 64):
            if n * n // sk >= 64:
                run("gram", m1, m1, n * n, 0, 1, sk, LOWER, 1.0)
                if sk >= 2:
                    run("gramxcd", m1, m1, n * n, 0, 1, sk, LOWER | XCD | NOFAST, 1.0)
    for sk in (1, 2, 4):
        run("chain", n, n, n, 1, 1, sk, 0, 0.0)
        run("chainT", n, n, n, 0, 1, sk, 0, 0.0)
        run("chainL", n, n, n, 0, 1, sk, LOWER, 0.0)
print("%d shapes differ" % bad)
