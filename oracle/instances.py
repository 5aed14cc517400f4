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
