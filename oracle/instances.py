"""instances.py - TEST INFRASTRUCTURE.  Seeded synthetic node SDPs with a planted strictly complementary optimum
(SURVEY.md section 8(d)): one dense block, no LP rows, free variables.

   A_i = (G_i + G_i^T) / sqrt(2 n),  G_i ~ N(0, 1)
   X* = Q diag(lx) Q^T (rank n/4),  Z* = Q diag(lz) Q^T (rank n - n/4),  X* Z* = 0,  eigenvalues U[1, 2]
   y* ~ U[-1, 1],   A_0 = sum_i A_i y*_i - Z*,   b_i = <A_i, X*>        ->   optimal value b^T y* = <A_0, X*>

The GPU bench generates the same instance on the device from the same counter-based stream (scip-sdp_amd/csrc/gen.hip);
tests/test_gen_parity.py checks that both produce identical bits.
"""
import numpy as np


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def counter_uniform(seed, idx):
    """U(0,1) doubles from a stateless 64-bit mix of (seed, idx); idx is a uint64 array."""
    with np.errstate(over='ignore'):
        h = splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + idx.astype(np.uint64))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def counter_normal(seed, idx):
    """N(0,1) by Box-Muller on two counter streams (cos branch only: one normal per index)."""
    u1 = counter_uniform(seed, 2 * idx.astype(np.uint64))
    u2 = counter_uniform(seed, 2 * idx.astype(np.uint64) + np.uint64(1))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def planted_dense(n, m, seed=20240, rank_frac=0.25):
    """returns (b[m], A[m+1, n, n], ystar, Xstar, Zstar)"""
    A = np.empty((m + 1, n, n))
    ii, jj = np.tril_indices(n)
    tri = (ii * (ii + 1) // 2 + jj).astype(np.uint64)
    L = n * (n + 1) // 2
    scale = 1.0 / np.sqrt(2.0 * n)
    for i in range(1, m + 1):
        g = counter_normal(seed + i, tri)             # lower triangle incl. diagonal, packed index order
        G = np.zeros((n, n))
        G[ii, jj] = g
        # (G + G^T) / sqrt(2n) with G lower triangular draws: off-diagonal g_ij, diagonal 2 g_ii
        A[i] = (G + G.T) * scale
    rng_idx = np.arange(n * n, dtype=np.uint64)
    Q, _ = np.linalg.qr(counter_normal(seed + 1000003, rng_idx).reshape(n, n))
    r = max(1, int(round(n * rank_frac)))
    ev = 1.0 + counter_uniform(seed + 2000003, np.arange(n, dtype=np.uint64))
    lx = np.where(np.arange(n) < r, ev, 0.0)
    lz = np.where(np.arange(n) < r, 0.0, ev)
    Xs = (Q * lx) @ Q.T
    Zs = (Q * lz) @ Q.T
    ys = 2.0 * counter_uniform(seed + 3000003, np.arange(m, dtype=np.uint64)) - 1.0
    A[0] = np.tensordot(ys, A[1:], axes=(0, 0)) - Zs
    A[0] = 0.5 * (A[0] + A[0].T)
    b = A[1:].reshape(m, -1) @ Xs.reshape(-1)
    return b, A, ys, Xs, Zs


def planted_sparse(n, m, nnz_per_matrix, seed=20240, rank_frac=0.25):
    """The same planted optimum with SPARSE constraint matrices: A_i (i >= 1) has nnz_per_matrix lower-triangular nonzeros at
    random positions (one of them on the diagonal), values N(0, 1) - what the reference's instances look like (1-10 nonzeros per
    matrix) and what both of its backends pass on as triplets (sdpisolver_dsdp.c:1126-1195, sdpisolver_sdpa.cpp:1223-1267).
    Returns (b[m], coo, A0[n, n], ystar, Xstar, Zstar) with coo = (var[1..m], row, col, val) of the variables' matrices,
    row >= col; the constant matrix A0 = sum_i A_i y*_i - Z* is dense."""
    rng = np.random.default_rng(seed)
    var, row, col, val = [], [], [], []
    for i in range(1, m + 1):
        seen = set()
        d = int(rng.integers(0, n))
        seen.add((d, d))
        while len(seen) < nnz_per_matrix:
            r, c = int(rng.integers(0, n)), int(rng.integers(0, n))
            seen.add((max(r, c), min(r, c)))
        for (r, c) in sorted(seen):
            var.append(i); row.append(r); col.append(c); val.append(float(rng.standard_normal()))
    var = np.array(var, dtype=np.int32); row = np.array(row, dtype=np.int32); col = np.array(col, dtype=np.int32)
    val = np.array(val)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    r = max(1, int(round(n * rank_frac)))
    ev = 1.0 + rng.random(n)
    Xs = (Q * np.where(np.arange(n) < r, ev, 0.0)) @ Q.T
    Zs = (Q * np.where(np.arange(n) < r, 0.0, ev)) @ Q.T
    ys = 2.0 * rng.random(m) - 1.0
    A0 = -Zs.copy()
    w = val * ys[var - 1]
    np.add.at(A0, (row, col), w)
    off = row != col
    np.add.at(A0, (col[off], row[off]), w[off])
    A0 = 0.5 * (A0 + A0.T)
    # b_i = <A_i, X*>: off-diagonal entries count twice
    contrib = val * Xs[row, col] * np.where(off, 2.0, 1.0)
    b = np.bincount(var - 1, weights=contrib, minlength=m)
    return b, (var, row, col, val), A0, ys, Xs, Zs


def coo_to_dense(n, m, coo, A0):
    """dense [m + 1, n, n] array of a planted_sparse instance (for the dense paths and the oracle)"""
    var, row, col, val = coo
    A = np.zeros((m + 1, n, n))
    A[0] = A0
    A[var, row, col] = val
    A[var, col, row] = val
    return A
