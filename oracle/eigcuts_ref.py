"""eigcuts_ref.py - TEST INFRASTRUCTURE.  Numpy restatement of the eigenvector-cut computation of the reference's LP-based
mode: cons_sdp.c:826-865 (multiplyConstraintMatrix: v^T A_j v from the lower-triangular entries, off-diagonal entries counted
twice), :896-952 (produceCutFromEigenvector: lhs = v^T A_0 v, coefficients v^T A_j v) and :1612-1803 (separateSol: the matrix
sum_j A_j y_j - A_0 at the point to separate, eigenvectors to its negative eigenvalues through
SCIPlapackComputeEigenvectorsNegative, lapack_interface.c:398-503)."""
import numpy as np


def vAv_sparse(entries, v):
    """cons_sdp.c:826-865 on lower-triangular (row, col, val) entries"""
    s = 0.0
    for (r, c, val) in entries:
        s += (1.0 if r == c else 2.0) * v[c] * val * v[r]
    return s


def cuts_dense(A, y, tol, maxcuts):
    """A[m + 1, n, n] with A[0] the constant matrix.  Returns (eigvals, coefs[k, m], lhs[k], vecs[k, n]) for the eigenvalues
    <= -tol of Z(y) = sum_i A[i] y_i - A[0], most negative first; cut: coefs @ y >= lhs."""
    m = A.shape[0] - 1
    Z = np.tensordot(y, A[1:], axes=(0, 0)) - A[0] if m > 0 else -A[0]
    w, V = np.linalg.eigh(Z)
    k = 0
    while k < len(w) and k < maxcuts and w[k] <= -tol:
        k += 1
    vecs = V[:, :k].T.copy()
    coefs = np.array([[float(v @ A[1 + i] @ v) for i in range(m)] for v in vecs]).reshape(k, m)
    lhs = np.array([float(v @ A[0] @ v) for v in vecs])
    return w[:k], coefs, lhs, vecs
