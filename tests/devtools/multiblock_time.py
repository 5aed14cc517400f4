#!/usr/bin/env python3
"""multiblock_time.py - developer tool: planted problems with SEVERAL dense blocks (K blocks of n rows, m variables): ms per IPM
iteration, time of the Schur assembly, and the rate of the assembly in algorithmic flops - to see what several small blocks cost
against one large one."""
import ctypes as C
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

hb = bench.load_binding()


def planted(K, n, m, seed):
    rng = np.random.default_rng(seed)
    y0 = rng.uniform(-1, 1, m)
    b = np.zeros(m)
    blocks = []
    for _ in range(K):
        G = rng.standard_normal((m + 1, n, n))
        A = (G + G.transpose(0, 2, 1)) / np.sqrt(2 * n)
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        r = max(1, n // 4)
        ev = rng.uniform(1, 2, n)
        Xs = (Q * np.where(np.arange(n) < r, ev, 0)) @ Q.T
        Zs = (Q * np.where(np.arange(n) < r, 0, ev)) @ Q.T
        A[0] = np.tensordot(y0, A[1:], axes=(0, 0)) - Zs
        b += A[1:].reshape(m, -1) @ Xs.reshape(-1)
        blocks.append(A)
    return types.SimpleNamespace(m=m, b=b, blocks=blocks, q=0, D=np.zeros((0, m)), c=np.zeros(0)), float(b @ y0)


for spec in (sys.argv[1:] or ["8,100,1000", "4,250,1000", "20,50,500", "2,500,1000", "1,500,1000", "16,64,1000", "3,130,300"]):
    K, n, m = (int(v) for v in spec.split(","))
    core, opt = planted(K, n, m, 7)
    s = hb.Solver(0)
    s.load_core(core)
    s.solve(gaptol=1e-5, feastol=1e-5)
    s.solve(gaptol=1e-5, feastol=1e-5)
    t0 = time.perf_counter()
    infos = [s.solve(gaptol=1e-5, feastol=1e-5) for _ in range(3)]
    el = (time.perf_counter() - t0) / 3
    i = infos[-1]
    alg = K * (4.0 * (m + 1) * n ** 3 + (m + 1) ** 2 * n ** 2)
    print("K %3d n %4d m %5d: %2d iterations, %8.3f ms/iter, assembly %8.3f ms (%5.1f TFLOP/s algorithmic), status %d, objective error %.1e" % (
        K, n, m, i.iterations, 1e3 * el / max(1, i.iterations), 1e3 * i.schur_seconds / max(1, i.schur_calls),
        alg / max(i.schur_seconds / max(1, i.schur_calls), 1e-12) / 1e12, i.status, abs(i.dobj - opt) / (1 + abs(opt))), flush=True)
    s.close()
