"""psd_project_ref.py - TEST INFRASTRUCTURE (oracle).  numpy restatement of the PSD projection chain of the warm-start producer,
src/scipsdp/relax_sdp.c:2715-2766 (dual matrix) / :3405-3445 (primal matrix) with the helpers expandSparseMatrix (:243-273) and
scaleTransposedMatrix (:276-302) and the two SCIPlapack calls it goes through (lapack_interface.c:507-603 DSYEVR 'A', :654-706
DGEMM).  Only tests/ may import this file.

Memory conventions restated literally: `eigenvectors` is the column-major Z of DSYEVR, i.e. as a flat array entry [k * n + j] is
component j of the k-th eigenvector; scaleTransposedMatrix multiplies flat entry [r * n + c] by scale[c]; the DGEMM call is
(A = eigenvectors, 'T', B = scaled, 'N') on column-major operands."""
import numpy as np

EPSILON = 1e-9          # SCIPepsilon: SCIPisLT(a, b) <=> a - b < -epsilon, SCIPisZero(x) <=> |x| <= epsilon


def expand_sparse(n, row, col, val):
    """relax_sdp.c:243-273"""
    full = np.zeros(n * n)
    for r, c, v in zip(row, col, val):
        full[r * n + c] = v
        full[c * n + r] = v
    return full


def chain(n, row, col, val, minev, eig=None, epsilon=EPSILON):
    """the literal chain; eig(full_flat) -> (eigenvalues ascending, flat eigenvector array) defaults to numpy's eigh.
    Returns (rows, cols, vals) of the entries r <= c with |value| > epsilon in the order the reference writes them, and the dense
    result."""
    full = expand_sparse(n, row, col, val)
    if eig is None:
        lam, U = np.linalg.eigh(full.reshape(n, n))
        vecs = np.ascontiguousarray(U.T).reshape(-1)          # flat [k * n + j] = component j of eigenvector k
    else:
        lam, vecs = eig(full)
        lam = np.array(lam, dtype=np.float64)
        vecs = np.array(vecs, dtype=np.float64).reshape(-1)
    lam = lam.copy()
    i = 0
    while i < n and lam[i] - minev < -epsilon:                # relax_sdp.c:2741-2742 (ascending order)
        lam[i] = minev
        i += 1
    scaled = vecs.copy()
    for r in range(n):                                        # scaleTransposedMatrix, relax_sdp.c:276-302
        scaled[r * n:(r + 1) * n] *= lam
    # DGEMM('T', 'N') on column-major views: A_cm[j, k] = vecs[k * n + j], B_cm likewise -> C_cm = A_cm^T B_cm; flat column-major
    A_cm = vecs.reshape(n, n).T
    B_cm = scaled.reshape(n, n).T
    C_cm = A_cm.T @ B_cm
    out = np.ascontiguousarray(C_cm.T).reshape(-1)            # column-major flat array of C
    rows, cols, vals = [], [], []
    for r in range(n):                                        # relax_sdp.c:2750-2763
        for c in range(r, n):
            v = out[r * n + c]
            if abs(v) > epsilon:
                rows.append(r); cols.append(c); vals.append(v)
    return np.array(rows, dtype=np.int32), np.array(cols, dtype=np.int32), np.array(vals), out.reshape(n, n)


def spectral(n, row, col, val, minev, epsilon=EPSILON):
    """sum_k max(lambda_k, minev) v_k v_k^T (what the comment at relax_sdp.c:2747 describes); basis independent"""
    full = expand_sparse(n, row, col, val).reshape(n, n)
    lam, U = np.linalg.eigh(full)
    lam = np.where(lam - minev < -epsilon, minev, lam)
    return (U * lam) @ U.T
