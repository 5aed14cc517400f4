"""shard_ref.py - TEST INFRASTRUCTURE.  Numpy restatement of the multi-GPU sharding of the Schur assembly
(scip-sdp_amd/csrc/schur.hip: hs_shard_rows / hs_schur_Urows, and the all-gather sequence in csrc/ipm.hip), used by the
world_size-2 gloo test.  Rank g of G computes two row chunks of the upper triangle, chunk g and chunk 2G-1-g of
c = ceil(m1 / 2G) rows each (equal triangle area per rank); chunks 0..G-1 are gathered in place, chunks G..2G-1 arrive in
reverse rank order and are copied to their rows; finally the upper triangle is mirrored."""
import numpy as np


def shard_rows(m1, nranks, rank):
    c = (m1 + 2 * nranks - 1) // (2 * nranks)
    return c, rank * c, (2 * nranks - 1 - rank) * c


def schur_rows(A, X, Zinv, r0, r1):
    """rows r0..r1 of Mx restricted to columns >= r0 (what one call of hs_schur_Urows adds)"""
    m1, n, _ = A.shape
    r1 = min(r1, m1)
    out = np.zeros((max(r1 - r0, 0), m1))
    if r1 <= r0:
        return out
    U = np.matmul(X, (A[r0:r1].reshape(-1, n) @ Zinv).reshape(-1, n, n))
    full = U.reshape(r1 - r0, -1) @ A[r0:].reshape(m1 - r0, -1).T
    out[:, r0:] = full
    for k in range(r1 - r0):          # only the upper triangle of these rows is defined (tiles left of the diagonal are skipped)
        out[k, :r0 + k] = 0.0
    return out


def local_contribution(A, X, Zinv, nranks, rank):
    m1 = A.shape[0]
    c, b1, b2 = shard_rows(m1, nranks, rank)
    return c, b1, b2, schur_rows(A, X, Zinv, b1, b1 + c), schur_rows(A, X, Zinv, b2, b2 + c)


def assemble(m1, nranks, c, first_chunks, second_chunks):
    """first_chunks[r], second_chunks[r]: the two row blocks of rank r (as gathered); returns the symmetric Mx"""
    Mx = np.zeros((m1 + 2 * nranks, m1))
    for r in range(nranks):
        blk = first_chunks[r]
        Mx[r * c:r * c + blk.shape[0]] = blk
        dst = (2 * nranks - 1 - r) * c
        blk2 = second_chunks[r]
        Mx[dst:dst + blk2.shape[0]] = blk2
    Mx = Mx[:m1]
    up = np.triu(Mx)
    return up + np.triu(Mx, 1).T
