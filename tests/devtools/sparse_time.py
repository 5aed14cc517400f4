#!/usr/bin/env python3
"""sparse_time.py [n m nnz_per_matrix] - developer tool: a planted problem whose matrices have a few nonzeros each (kept as triplets by the
engine, csrc/sparse.hip): ms per iteration; run under tools/prof_any.sh for the kernel statistics."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

hb = bench.load_binding()
n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 500
m2 = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
seed = 20240
rng = np.random.default_rng(seed)
var = np.repeat(np.arange(1, m2 + 1, dtype=np.int32), k)
r = rng.integers(0, n2, size=m2 * k)
c = rng.integers(0, n2, size=m2 * k)
c[::k] = r[::k]
row, col = np.maximum(r, c).astype(np.int32), np.minimum(r, c).astype(np.int32)
key = var.astype(np.int64) * n2 * n2 + row.astype(np.int64) * n2 + col
_, first = np.unique(key, return_index=True)
var, row, col = var[first], row[first], col[first]
val = rng.standard_normal(len(var))
Xs, Zs, ys = bench.planted_pair(n2, m2, seed + 1)
A0 = -Zs.copy()
w = val * ys[var - 1]
np.add.at(A0, (row, col), w)
off = row != col
np.add.at(A0, (col[off], row[off]), w[off])
A0 = 0.5 * (A0 + A0.T)
b = np.bincount(var - 1, weights=val * Xs[row, col] * np.where(off, 2.0, 1.0), minlength=m2)
s = hb.Solver(0)
s.load_sparse(m2, n2, b, (var, row, col, val), A0)
print("kept as nonzeros:", s.is_sparse(0))
for _ in range(2):
    s.solve(gaptol=1e-5, feastol=1e-5)
t0 = time.perf_counter()
infos = [s.solve(gaptol=1e-5, feastol=1e-5) for _ in range(3)]
el = (time.perf_counter() - t0) / 3
i = infos[-1]
print("n %d m %d, %d nonzeros per matrix: %d iterations, %.3f ms per iteration, assembly %.3f ms, status %d, objective error %.1e" % (
    n2, m2, k, i.iterations, 1e3 * el / i.iterations, 1e3 * i.schur_seconds / max(1, i.schur_calls), i.status, abs(i.dobj - float(b @ ys)) / (1 + abs(float(b @ ys)))))
s.close()
