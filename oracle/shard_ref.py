"""shard_ref.py - TEST INFRASTRUCTURE.  Numpy restatement of the multi-GPU sharding of the Schur assembly
(scip-sdp_amd/csrc/schur.hip: hs_shard_rows / hs_schur_Urows, and the all-gather sequence in csrc/ipm.hip), used by the
world_size-2 gloo test.  Rank g of G computes two row chunks of the upper triangle, chunk g and chunk 2G-1-g of
c = ceil(m1 / 2G) rows each (equal triangle area per rank); chunks 0..G-1 are gathered in place, chunks G..2G-1 arrive in
reverse rank order and are copied to their rows; finally the upper triangle is mirrored.

Second form (the default of the engine with several ranks; hs_shard_cols / hs_schur_Wcols + one all-reduce): with X = R R^T,
Z^-1 = G^T G and W_j = G A_j R, Mx = sum over the entries (r, c) of the W_j; the rank that owns the columns [c0, c0 + cw)
adds W[:, :, c0:c0+cw] flattened times its transpose, and the partial matrices are summed over the ranks.

Third form (matrices sharded by variable, hipsdp_shard_matrices; hs_var_rows / hs_var_wrows / hs_schur_Wvar + all-to-all +
all-reduce): rank g holds A_j only for its own rows j in [g c, (g + 1) c), c = ceil(m1 / G), forms the column slice of its
W_j, cuts the rows of the slice into G ranges [n h / G, n (h + 1) / G) and sends range h to rank h; rank h then holds its row
range of ALL W_j and adds that part of the Gram matrix."""
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


def shard_cols(m1, n, nranks):
    """restatement of hs_shard_cols: boundaries at multiples of 16, minimal slowest rank under the cost model of schur.hip
    (width rounded up to the 128-wide tile - 64 for slices of at most 64 columns - in the two triangular products, plain width in W W^T)"""
    gran = 16
    P = (n + gran - 1) // gran
    col = lambda p: min(p * gran, n)

    def cost(p0, p1):
        c0, w = col(p0), col(p1) - col(p0)
        if w <= 0:
            return 0.0
        wt = 64.0 if w <= 64 else 128.0 * ((w + 127) // 128)      # slices of <= 64 columns run on 64-wide tiles
        return wt * (2.0 * (n - c0 - 0.5 * w) + n) + float(w) * m1

    prev = [1e300] * (P + 1)
    prev[0] = 0.0
    frm = [[0] * (P + 1) for _ in range(nranks + 1)]
    for g in range(1, nranks + 1):
        cur = [1e300] * (P + 1)
        for p in range(P + 1):
            b, arg = 1e300, 0
            for q in range(p + 1):
                if prev[q] >= 1e300:
                    continue
                v = max(prev[q], cost(q, p))
                if v < b:
                    b, arg = v, q
            cur[p], frm[g][p] = b, arg
        prev = cur
    bounds = [0] * (nranks + 1)
    bounds[nranks] = n
    p = P
    for g in range(nranks, 0, -1):
        p = frm[g][p]
        bounds[g - 1] = col(p)
    return bounds


def column_slice_contribution(A, R, G, c0, cw):
    """what hs_schur_Wcols adds for the columns [c0, c0 + cw): W_slice W_slice^T with W_j = G A_j R"""
    m1, n, _ = A.shape
    if cw <= 0:
        return np.zeros((m1, m1))
    T = A[:, :, c0:] @ R[c0:, c0:c0 + cw]           # R lower triangular: rows < c0 of these columns are zero
    W = np.matmul(G, T).reshape(m1, -1)
    return W @ W.T


def var_rows(m1, nranks, rank):
    """rows of A (0 = constant matrix, i = variable i) rank holds: hs_var_rows, also the row split of the sharded passes"""
    c = (m1 + nranks - 1) // nranks
    r0 = min(rank * c, m1)
    return r0, min(r0 + c, m1)


def var_wrows(n, nranks, rank):
    """rows of the W_j a rank receives in the all-to-all: hs_var_wrows"""
    return n * rank // nranks, n * (rank + 1) // nranks


def var_send_pieces(A_own, R, G, nranks, c0, cw):
    """what hs_schur_Wvar puts into the send buffer for the column slice [c0, c0 + cw): for every destination h the rows
    [q0_h, q1_h) of W_j[:, c0:c0+cw], j over the own rows, as one array [own rows, q1 - q0, cw]"""
    n = R.shape[0]
    W = np.matmul(G, np.matmul(A_own, R[:, c0:c0 + cw])) if A_own.shape[0] else np.zeros((0, n, cw))
    return [np.ascontiguousarray(W[:, slice(*var_wrows(n, nranks, h)), :]) for h in range(nranks)]


def var_gram(received):
    """received[src]: [rows of src, my row range, cw] in source order -> my part of W W^T for this slice"""
    Wt = np.concatenate([r.reshape(r.shape[0], r.shape[1] * r.shape[2]) for r in received], axis=0)
    return Wt @ Wt.T
