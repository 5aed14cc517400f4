"""GPU parity tests of the single kernels, through the C ABI (hipsdp_* host-buffer entry points), against numpy/LAPACK on
the same seeded inputs.  FP64 everywhere; tolerances are relative to the result's magnitude and stated per test."""
import numpy as np
import pytest

import ipm_ref

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(20240)


def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 53, 29), (130, 257, 100), (300, 300, 1111), (1500, 140, 70)])
@pytest.mark.parametrize("layA,layB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_dgemm_all_layouts(gpu, M, N, K, layA, layB):
    A = RNG.standard_normal((M, K) if layA == 0 else (K, M))
    B = RNG.standard_normal((N, K) if layB == 0 else (K, N))
    C0 = RNG.standard_normal((M, N))
    ref = 0.75 * (A if layA == 0 else A.T) @ (B.T if layB == 0 else B) - 0.5 * C0
    out = gpu.dgemm(A, B, layA, layB, alpha=0.75, beta=-0.5, Cin=C0)
    assert rel(out, ref) <= 1e-13 * max(1, K) ** 0.5


def test_dgemm_splitk_is_deterministic_and_lower_only(gpu):
    A = RNG.standard_normal((520, 4000))
    B = RNG.standard_normal((520, 4000))
    ref = A @ B.T
    o1 = gpu.dgemm(A, B, 0, 0, lower_only=True, splitk=5)
    o2 = gpu.dgemm(A, B, 0, 0, lower_only=True, splitk=5)
    assert np.array_equal(np.tril(o1), np.tril(o2))                 # slice-ordered reduction: bitwise reproducible
    assert rel(np.tril(o1), np.tril(ref)) <= 1e-12


def test_checklapack_golden_gemm(gpu):
    """unittests/src/checklapack.c:80-119 through the device GEMM: column-major [1 3;2 4] * [5 7;6 8]^T = [26 30;38 44]"""
    Acm = np.array([1, 2, 3, 4], dtype=float)     # column-major 2x2
    Bcm = np.array([5, 6, 7, 8], dtype=float)
    # column-major X[M x K] is the row-major array [K][M]: layout HS_MC for A; B^T with B column-major [N x K] -> [K][N]
    out = gpu.dgemm(Acm.reshape(2, 2), Bcm.reshape(2, 2), layA=1, layB=1)
    assert np.allclose(out.T.reshape(-1), [26, 38, 30, 44])


@pytest.mark.parametrize("m1,n", [(4, 3), (9, 5), (41, 20), (101, 50), (150, 64), (33, 130)])
def test_schur_assembly(gpu, m1, n):
    A = RNG.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    G = RNG.standard_normal((n, n)); X = G @ G.T + np.eye(n)
    G = RNG.standard_normal((n, n)); Zi = np.linalg.inv(G @ G.T + np.eye(n))
    ref = ipm_ref.schur_block(A, X, Zi)
    assert rel(gpu.schur_dense(A, X, Zi), ref) <= 1e-12
    assert rel(gpu.schur_dense(A, X, Zi, ws_gbytes=16.0 * n * n * 3 / 1e9), ref) <= 1e-12     # chunked over variables


@pytest.mark.parametrize("m1,n", [(4, 3), (41, 20), (150, 64), (33, 130), (300, 140)])
def test_schur_assembly_w_formulation(gpu, m1, n):
    """W_j = G A_j R, Mx = W W^T (triangular-aware GEMMs + XCD-sliced SYRK) against the direct formula"""
    A = RNG.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    G = RNG.standard_normal((n, n)); X = G @ G.T + np.eye(n)
    G = RNG.standard_normal((n, n)); Z = G @ G.T + np.eye(n)
    ref = ipm_ref.schur_block(A, X, np.linalg.inv(Z))
    assert rel(gpu.schur_w(A, X, Z), ref) <= 1e-11


@pytest.mark.parametrize("n", [1, 2, 7, 63, 64, 65, 130, 192, 300, 777, 1000, 1601])
def test_cholesky_inverse_and_solves(gpu, n):
    G = RNG.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    L, fail = gpu.potrf(S)
    assert fail == 0
    assert rel(L, np.linalg.cholesky(S)) <= 1e-13
    assert rel(gpu.trtri(S), np.linalg.inv(np.linalg.cholesky(S))) <= 1e-12
    r = RNG.standard_normal((3, n))
    assert rel(gpu.potrs(S, r), np.linalg.solve(S, r.T).T) <= 1e-11


@pytest.mark.parametrize("n", [40, 64, 130, 300, 1000])
def test_corrected_solves_on_an_ill_conditioned_factor(gpu, n):
    """hipsdp_potrs runs the triangular solves as the engine does: block by block through the explicit inverses of the 64 x 64
    diagonal blocks, each solve corrected once with the factor itself (hs_trsv mode bit 4; the single-workgroup kernel for
    n <= 128, the multi-workgroup kernels above).  Eigenvalues from 1 down to 1e-12: the residual of M x = b stays at the level
    LAPACK's substitutions leave.  (Norm-wise this also holds without the correction; what the correction buys shows in the
    interior-point iteration, tests/test_gpu_sdpi_branches.py::test_node_without_attained_optimum_...)"""
    import scipy.linalg as sla
    Q, _ = np.linalg.qr(RNG.standard_normal((n, n)))
    lam = 10.0 ** (-12.0 * np.arange(n) / max(n - 1, 1))
    M = (Q * lam) @ Q.T
    M = 0.5 * (M + M.T)
    b = M @ RNG.standard_normal((2, n)).T
    x = gpu.potrs(M, b.T.copy()).T
    c, low = sla.cho_factor(M, lower=True)
    xr = sla.cho_solve((c, low), b)
    res = np.linalg.norm(M @ x - b) / np.linalg.norm(b)
    res_ref = np.linalg.norm(M @ xr - b) / np.linalg.norm(b)
    assert res <= 20.0 * res_ref + 1e-15, (res, res_ref)


@pytest.mark.parametrize("n", [65, 128, 130, 192, 500, 1000, 1001])
def test_fused_block_column_cholesky_matches_the_four_launch_form_bitwise(gpu, n):
    """k_potrf_step (one launch per block column: narrow update, diagonal factorization, panel product, mask and the trailing
    update of the previous column) performs the same operations in the same order per element as diagonal kernel + panel GEMM +
    mask kernel + trailing GEMM: identical bits in L, in the inverses of the diagonal blocks, in the forced-pivot mask - definite
    and semidefinite mode (rank-deficient Schur matrices are where the pivot rule matters)"""
    G = RNG.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    L0, d0, _, f0 = gpu.potrf_ex(S, psd=False, v1=True)
    L1, d1, _, f1 = gpu.potrf_ex(S, psd=False, v1=False)
    assert f0 == 0 and f1 == 0
    assert np.array_equal(np.tril(L0), np.tril(L1)) and np.array_equal(d0, d1)
    assert rel(np.tril(L1), np.linalg.cholesky(S)) <= 1e-13
    # semidefinite: rank n // 2 + 3 Gram matrix, scaled rows
    r = n // 2 + 3
    H = RNG.standard_normal((n, r))
    P = H @ H.T
    L0, d0, m0, _ = gpu.potrf_ex(P, psd=True, v1=True)
    L1, d1, m1, _ = gpu.potrf_ex(P, psd=True, v1=False)
    assert np.array_equal(m0, m1) and m1.sum() > 0
    assert np.array_equal(np.tril(L0), np.tril(L1)) and np.array_equal(d0, d1)
    Lt = np.tril(L1)
    assert rel(Lt @ Lt.T, P) <= 1e-9                     # forced pivots are tiny: the factor still reproduces the matrix
    # an indefinite matrix is flagged at the same pivot
    S2 = S.copy()
    S2[n - 3, n - 3] = -1.0
    assert gpu.potrf_ex(S2, v1=True)[3] == gpu.potrf_ex(S2, v1=False)[3] != 0


def test_cholesky_flags_indefinite_matrix(gpu):
    S = np.eye(70)
    S[40, 40] = -1.0
    L, fail = gpu.potrf(S)
    assert fail == 41


@pytest.mark.parametrize("n", [1, 2, 10, 43, 128, 400])
def test_lambda_min_lanczos(gpu, n):
    G = RNG.standard_normal((n, n))
    W = G + G.T
    ev = np.linalg.eigvalsh(W)
    theta, resid = gpu.lambda_min(W, 0)        # min(n, 250) steps
    assert abs(theta - ev[0]) <= 1e-9 * max(1.0, abs(ev[0])) + 2 * resid
    theta2, resid2 = gpu.lambda_min(W, 24)     # the interior-point setting: estimate + bound
    assert theta2 >= ev[0] - 1e-9 * abs(ev[0])                      # a Ritz value never undershoots
    assert theta2 - resid2 <= ev[0] + 0.35 * abs(ev[0])             # and the pessimistic value is not wildly off


@pytest.mark.parametrize("n", [1, 2, 3, 10, 43, 64, 65, 97, 128, 300, 513, 1000])
def test_syev_matches_dsyevr_semantics(gpu, n):
    """ascending eigenvalues, eigenvectors as ROWS (lapack_interface.c:507-603); n > 64 runs the block Jacobi (32 x 32 blocks,
    64 x 64 pair subproblems in LDS, rotations applied as 64-deep products), 65 / 97 / 300 / 513 / 1000 need zero padding"""
    G = RNG.standard_normal((n, n))
    W = G + G.T
    lam, V = gpu.syev(W)
    ev = np.linalg.eigvalsh(W)
    assert rel(lam, ev) <= 2e-12
    assert np.abs(V @ V.T - np.eye(n)).max() <= 2e-12
    assert np.abs(V @ W @ V.T - np.diag(lam)).max() <= 1e-11 * max(1.0, np.abs(ev).max())


@pytest.mark.parametrize("n", [2, 3, 5, 8, 16, 17, 31, 32, 33, 50, 63, 64])
def test_small_full_decomposition_in_one_launch(gpu, n):
    """n <= 64: tridiagonal reduction, all eigenvalues by multisection, eigenvectors by inverse iteration with re-orthogonalisation
    inside clusters, back-transformation - one launch (csrc/eigi.hip, what DSYEVR RANGE = 'A' does: lapack_interface.c:507-603).
    The spectra a PSD projection meets: random, diagonal input (split tridiagonal matrix), multiples of the identity (one cluster
    of n equal eigenvalues), low rank (a cluster at zero), pairs closer than the cluster tolerance, graded over 12 orders."""
    rng = np.random.default_rng(100 + n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    cases = {"random": (lambda G: G + G.T)(rng.standard_normal((n, n))),
             "diagonal": np.diag(rng.standard_normal(n)),
             "identity": 3.5 * np.eye(n),
             "low_rank": (lambda B: B @ B.T)(rng.standard_normal((n, max(1, n // 4)))),
             "close_pairs": (Q * np.repeat(np.arange(1, n // 2 + 2, dtype=float), 2)[:n] * (1 + 1e-9 * np.arange(n))) @ Q.T,
             "graded": (Q * 10.0 ** np.linspace(-6, 6, n)) @ Q.T,
             "two_clusters": (Q * np.where(np.arange(n) < n // 2, -1.0, 2.0)) @ Q.T}
    for name, W in cases.items():
        W = 0.5 * (W + W.T)
        lam, V = gpu.syev(W)
        ev = np.linalg.eigvalsh(W)
        scale = max(1.0, np.abs(ev).max())
        assert np.abs(lam - ev).max() <= 1e-12 * scale, (name, np.abs(lam - ev).max())
        assert np.all(np.diff(lam) >= 0.0), name
        assert np.abs(V @ V.T - np.eye(n)).max() <= 1e-11, (name, np.abs(V @ V.T - np.eye(n)).max())
        assert np.abs(V @ W @ V.T - np.diag(lam)).max() <= 1e-11 * scale, (name, np.abs(V @ W @ V.T - np.diag(lam)).max())


@pytest.mark.parametrize("n", [65, 66, 80, 100, 127, 128])
def test_mid_full_decomposition_in_one_launch(gpu, n):
    """64 < n <= 128: the same in one launch with the matrix in LDS (k_syev_mid: reflectors and factor arrays in device memory, the
    eigenvectors of the tridiagonal matrix in the LDS the matrix leaves) - the spectra of the test above."""
    rng = np.random.default_rng(100 + n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    cases = {"random": (lambda G: G + G.T)(rng.standard_normal((n, n))),
             "diagonal": np.diag(rng.standard_normal(n)),
             "identity": 3.5 * np.eye(n),
             "low_rank": (lambda B: B @ B.T)(rng.standard_normal((n, max(1, n // 4)))),
             "close_pairs": (Q * np.repeat(np.arange(1, n // 2 + 2, dtype=float), 2)[:n] * (1 + 1e-9 * np.arange(n))) @ Q.T,
             "graded": (Q * 10.0 ** np.linspace(-6, 6, n)) @ Q.T,
             "two_clusters": (Q * np.where(np.arange(n) < n // 2, -1.0, 2.0)) @ Q.T}
    for name, W in cases.items():
        W = 0.5 * (W + W.T)
        lam, V = gpu.syev(W)
        ev = np.linalg.eigvalsh(W)
        scale = max(1.0, np.abs(ev).max())
        assert np.abs(lam - ev).max() <= 1e-12 * scale, (name, np.abs(lam - ev).max())
        assert np.all(np.diff(lam) >= 0.0), name
        assert np.abs(V @ V.T - np.eye(n)).max() <= 1e-11, (name, np.abs(V @ V.T - np.eye(n)).max())
        assert np.abs(V @ W @ V.T - np.diag(lam)).max() <= 1e-11 * scale, (name, np.abs(V @ W @ V.T - np.diag(lam)).max())


@pytest.mark.parametrize("n", [130, 200, 300, 500])
def test_block_jacobi_on_clustered_spectra(gpu, n):
    """n > 128 (block Jacobi, csrc/eig.hip: hs_syev_jacobi) on spectra with multiple eigenvalues - a low-rank matrix shifted by a
    constant is what the PSD projection of a warm start meets (relax_sdp.c:2733-2766).  With the n - rank equal eigenvalues scattered
    over the diagonal the cyclic method converges linearly at the end (the decomposition of n = 500 / rank 50 stopped at the sweep
    limit with 7e-12 left off the diagonal); the driver sorts the coordinates by their diagonal entries when a sweep does not bring
    the quadratic drop.  Same accuracy as for separated spectra."""
    rng = np.random.default_rng(300 + n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    cases = {"low_rank_shifted": (lambda B: B @ B.T - 0.01 * np.eye(n))(rng.standard_normal((n, n // 10))),
             "rank_one": (lambda b: np.outer(b, b))(rng.standard_normal(n)),
             "two_clusters": (Q * np.where(np.arange(n) < n // 2, -1.0, 2.0)) @ Q.T,
             "identity": 3.5 * np.eye(n),
             "random": (lambda G: G + G.T)(rng.standard_normal((n, n)))}
    for name, W in cases.items():
        W = 0.5 * (W + W.T)
        lam, V = gpu.syev(W)
        ev = np.linalg.eigvalsh(W)
        scale = max(1.0, np.abs(ev).max())
        assert np.abs(lam - ev).max() <= 1e-12 * scale, (name, np.abs(lam - ev).max())
        assert np.all(np.diff(lam) >= 0.0), name
        assert np.abs(V @ V.T - np.eye(n)).max() <= 1e-11, (name, np.abs(V @ V.T - np.eye(n)).max())
        assert np.abs(V @ W @ V.T - np.diag(lam)).max() <= 2e-13 * scale, (name, np.abs(V @ W @ V.T - np.diag(lam)).max())


@pytest.mark.parametrize("R,E", [(1, 1), (5, 100), (37, 10000), (300, 40001), (1001, 2500)])
def test_gemv_passes(gpu, R, E):
    A = RNG.standard_normal((R, E))
    V = RNG.standard_normal((3, E))
    c = RNG.standard_normal(R)
    assert rel(gpu.gemv_n(A, V), V @ A.T) <= 1e-13 * E ** 0.5
    assert rel(gpu.gemv_t(A, c), c @ A) <= 1e-13 * R ** 0.5


# ---- the SCIPlapack* surface (include/lapack_interface_hip.h) -----------------------------------------------------------
import ctypes as C


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_sciplapack_matrix_matrix_mult_checklapack(gpu):
    """unittests/src/checklapack.c:80-119 verbatim: A = {1,2,3,4}, B = {5,6,7,8} column-major, transposeB = TRUE"""
    lib = gpu.lib()
    A = np.array([1, 2, 3, 4], dtype=float)
    B = np.array([5, 6, 7, 8], dtype=float)
    out = np.zeros(4)
    rc = lib.SCIPlapackMatrixMatrixMult(2, 2, _pd(A), 0, 2, 2, _pd(B), 1, _pd(out))
    assert rc == 1
    assert np.allclose(out, [26, 38, 30, 44])
    # all four transpose combinations on rectangular data against numpy (column-major views)
    rng = np.random.default_rng(5)
    for ta in (0, 1):
        for tb in (0, 1):
            M, N, K = 5, 7, 4
            Af = rng.standard_normal((K, M) if ta else (M, K))
            Bf = rng.standard_normal((N, K) if tb else (K, N))
            ref = (Af.T if ta else Af) @ (Bf.T if tb else Bf)
            a = np.asfortranarray(Af).reshape(-1, order="F").copy()
            b = np.asfortranarray(Bf).reshape(-1, order="F").copy()
            o = np.zeros(M * N)
            rc = lib.SCIPlapackMatrixMatrixMult(Af.shape[0], Af.shape[1], _pd(a), ta, Bf.shape[0], Bf.shape[1], _pd(b), tb, _pd(o))
            assert rc == 1
            assert np.allclose(o.reshape(N, M).T, ref, atol=1e-13)


def test_sciplapack_eigen_and_gemv(gpu):
    lib = gpu.lib()
    rng = np.random.default_rng(6)
    n = 12
    G = rng.standard_normal((n, n))
    W = G + G.T
    ev, U = np.linalg.eigh(W)
    val = C.c_double(0)
    vec = np.zeros(n)
    for i in (1, 3, n):
        a = W.copy().reshape(-1)
        assert lib.SCIPlapackComputeIthEigenvalue(None, 1, n, _pd(a), i, C.byref(val), _pd(vec)) == 1
        assert abs(val.value - ev[i - 1]) <= 1e-11
        assert np.allclose(W @ vec, val.value * vec, atol=1e-10) and abs(np.linalg.norm(vec) - 1) <= 1e-12
    lam = np.zeros(n)
    V = np.zeros(n * n)
    a = W.copy().reshape(-1)
    assert lib.SCIPlapackComputeEigenvectorDecomposition(None, n, _pd(a), _pd(lam), _pd(V)) == 1
    V = V.reshape(n, n)
    assert np.allclose(lam, ev, atol=1e-11) and np.allclose(V.T @ np.diag(lam) @ V, W, atol=1e-10)   # rows are eigenvectors
    cnt = C.c_int(0)
    a = W.copy().reshape(-1)
    assert lib.SCIPlapackComputeEigenvectorsNegative(None, n, _pd(a), C.c_double(1e-6), C.byref(cnt), _pd(lam), _pd(V.reshape(-1))) == 1
    assert cnt.value == int(np.sum(ev <= -1e-6)) and np.allclose(lam[:cnt.value], ev[:cnt.value], atol=1e-11)
    Mx = rng.standard_normal((5, 3))
    x = rng.standard_normal(3)
    y = np.zeros(5)
    m_cm = Mx.reshape(-1, order="F").copy()
    assert lib.SCIPlapackMatrixVectorMult(5, 3, _pd(m_cm), _pd(x), _pd(y)) == 1
    assert np.allclose(y, Mx @ x, atol=1e-13)
    # least squares, rank deficient
    A = rng.standard_normal((6, 4)); A[:, 3] = A[:, 0] + A[:, 1]
    b = rng.standard_normal(6)
    xs = np.zeros(4)
    a_cm = A.reshape(-1, order="F").copy()
    assert lib.SCIPlapackLinearSolve(None, 6, 4, _pd(a_cm), _pd(b.copy()), _pd(xs)) == 1
    assert np.allclose(xs, np.linalg.lstsq(A, b, rcond=None)[0], atol=1e-8)
    # an ill-conditioned system (cond 1e5: the normal equations alone lose 10 of 16 digits): the refinement steps bring the
    # answer to what DGELSD's SVD gives
    rng2 = np.random.default_rng(11)
    U, _ = np.linalg.qr(rng2.standard_normal((12, 5)))
    W, _ = np.linalg.qr(rng2.standard_normal((5, 5)))
    A2 = (U * np.array([1.0, 1e-1, 1e-2, 1e-4, 1e-5])) @ W.T
    b2 = A2 @ rng2.standard_normal(5) + 1e-3 * rng2.standard_normal(12)
    x2 = np.zeros(5)
    a2_cm = np.asfortranarray(A2).reshape(-1, order="F").copy()
    assert lib.SCIPlapackLinearSolve(None, 12, 5, _pd(a2_cm), _pd(b2.copy()), _pd(x2)) == 1
    xr = np.linalg.lstsq(A2, b2, rcond=None)[0]
    assert np.linalg.norm(x2 - xr) <= 1e-6 * np.linalg.norm(xr)


# flags of hs_gemm_args (csrc/hs_common.h)
GEMM_LOWER, GEMM_A_LOWTRI, GEMM_B_LOWTRI, GEMM_XCD, GEMM_REMAP, GEMM_NOFAST = 1, 2, 4, 8, 16, 64


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,layB,batch,splitk,flags,beta,expect_v2", [
    (50000, 500, 500, 1, 1, 1, 0, 0.0, 1),                              # stack product without the triangular clipping
    (1001, 1001, 30000, 0, 1, 14, GEMM_LOWER | GEMM_XCD | GEMM_NOFAST, 1.0, 1),   # Gram product W W^T in K slices, ragged last tile
    (3000, 2500, 500, 1, 1, 1, 0, 0.5, 1),                              # beta != 0
    (2900, 2600, 330, 0, 1, 1, 0, 0.0, 1),                              # both K contiguous, ragged tiles, K tail of 2
    (3001, 2001, 402, 0, 1, 1, 0, 0.0, 1),                              # odd M and N
    (3000, 2502, 128, 1, 1, 1, 0, 0.0, 1),                              # row-contiguous B, ragged last column tile
    (3000, 2500, 501, 1, 1, 1, 0, 0.0, 0),                              # odd K: not eligible, both runs take the tile kernel
    (401, 401, 25000, 0, 1, 64, GEMM_LOWER, 1.0, 1),                    # 64 slices asked, 63 exist after rounding the slice length: no empty slab is reduced
    (401, 401, 10000, 0, 1, 39, GEMM_LOWER, 1.0, 0),                    # 37 of 39 slices exist (too few items for the persistent kernel)
    (700, 700, 20000, 0, 1, 19, GEMM_LOWER | GEMM_XCD | GEMM_NOFAST, 1.0, 1),     # XCD-walked slices with an empty tail (1056 per slice)
])
def test_persistent_gemm_matches_tile_gemm_bitwise(gpu, M, N, K, layB, batch, splitk, flags, beta, expect_v2):
    used, ndiff = gpu.dgemm_selfcheck(M, N, K, layB=layB, batch=batch, splitk=splitk, flags=flags, beta=beta)
    assert used == expect_v2
    assert ndiff == 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,layB,batch,flags,alpha,beta", [
    (50000, 500, 500, 1, 1, GEMM_B_LOWTRI, 1.25, 0.0),                  # stack product of the Schur assembly (A_j R): last column tile 116 wide, K tail of 4
    (50000, 500, 500, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),                   # alpha = 1: the store path the assembly takes
    (500, 500, 500, 1, 40, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),       # batched product G T_j, T_j row-contiguous
    (500, 500, 500, 0, 40, GEMM_A_LOWTRI | GEMM_REMAP, 1.25, 0.0),      # T_j given K contiguous (transposed storage)
    (40000, 256, 384, 1, 1, GEMM_B_LOWTRI, 1.0, 0.5),                   # K > N (column slice of a sharded assembly), beta != 0
    (50000, 372, 500, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),                   # ragged last column tile, the band of the last tile cut by K
    (30001, 1000, 1000, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),                 # eight column tiles, odd M (clamped rows)
    (1000, 1000, 1000, 1, 16, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),    # eight row tiles per matrix: up to 112 full stages in front of the band
    (362, 362, 362, 1, 160, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),      # three row tiles, the last 106 rows: band cut by K
    (6000, 130, 130, 1, 6, GEMM_B_LOWTRI, 1.0, 0.0),                    # batched right-triangular product, second column tile 2 wide
    # the list order in sets of four panels of the triangular operand (g5_decode_sets): 5, 6, 7 and 9 panels leave last sets of 1, 2, 3, 1
    (20000, 640, 640, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),
    (20000, 700, 700, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),
    (20000, 896, 896, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),
    (12000, 1100, 1100, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),
    (640, 640, 640, 1, 24, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),
    (760, 760, 760, 1, 16, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),
    (896, 896, 896, 1, 12, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),
    (1100, 1100, 1100, 1, 9, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),
    (6000, 2048, 2048, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),                  # sixteen panels: four whole sets (n = 2000 is BASELINE configs[3])
    (2048, 2048, 2048, 1, 3, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),
    (3000, 4100, 4100, 1, 1, GEMM_B_LOWTRI, 1.0, 0.0),                  # 33 panels, the last set a single one
    (1000, 1000, 1000, 1, 7, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),     # 7 x 64 items in whole batch entries: the eighth XCD's share is empty
    (100, 100, 100, 1, 500, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),      # one panel
    (250, 250, 250, 1, 120, GEMM_A_LOWTRI | GEMM_REMAP, 1.0, 0.0),      # two panels, lists shorter than the re-ordered tail
])
def test_paired_band_gemm_matches_tile_gemm(gpu, M, N, K, layB, batch, flags, alpha, beta):
    """the two triangular n^3 products through the paired-band kernel (csrc/dgemm2.hip: hs_dgemm5_kernel - double stages (d, 15 - d)
    inside the diagonal band, list entries taken dynamically) against the one-tile-per-workgroup kernel on the same
    device-generated operands.  The band is summed in another order of K, so the results agree to rounding, not bit for bit:
    entries are sums of at most K products of numbers in [-0.5, 0.5), |difference| <= 1e-15 K is 5-10 ulp of the largest of them.
    A second call must reproduce the first bit for bit (which workgroup computes a tile must not matter)."""
    used, ndiff, maxdiff, nrepro, _, _ = gpu.dgemm_selfcheck3(M, N, K, layB=layB, batch=batch, flags=flags, alpha=alpha, beta=beta)
    assert used == 1
    assert nrepro == 0
    assert maxdiff <= 1e-15 * K
    assert ndiff > 0 or K <= 136           # (really another summation order: the test would otherwise not exercise the new kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,expect_gram", [
    (1001, 120000, 1),       # the tile count of the bench shape (36 lower tiles, ragged last tile): 15 / 11 K slices, one item per workgroup
    (401, 250000, None),     # 4 x 4 tiles: the diagonal tiles are 40 % of the items
    (700, 120000, None),
    (1024, 65536, None),     # whole tiles
    (1001, 30000, None),     # short K: taken only if the plan's estimate beats the tile kernel's by 3 % (csrc/gram.hip: gr_plan)
    (2001, 20000, 0),        # 136 tiles: more than one item per workgroup - left to the K-sliced tile kernel
    (200, 40000, 0),         # below the sizes the kernel is offered
])
def test_gram_kernel_matches_the_k_sliced_tile_kernels(gpu, M, K, expect_gram):
    """Mx += W W^T on the lower triangle through csrc/gram.hip (diagonal tiles with nine accumulators per wavefront from one operand
    image, tiles cut into K slices by a plan made on the host, one item per workgroup) against the K-sliced tile kernels on the same
    device-generated W: the two sum K in different slices, so they agree to rounding (entries are sums of K products of numbers in
    [-0.5, 0.5): the diagonal is about K / 12); a second run of the Gram kernel reproduces the first bit for bit"""
    used, maxdiff, nrepro, _, _ = gpu.gram_selfcheck(M, K)
    assert expect_gram is None or used == expect_gram
    if used:
        assert nrepro == 0
        assert maxdiff <= 1e-15 * K


@pytest.mark.gpu
def test_schur_products_at_bench_shape_against_numpy(gpu):
    """the two dominant product shapes of the n = 500, m = 1000 assembly through the production dispatch against numpy (the
    kernels are otherwise compared with each other at this size): the stack product A_stack R (500500 x 500 x 500) and the Gram
    product W W^T (1001 x 1001 x 250000, lower tiles, K slices)"""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((500500, 500))
    R = rng.standard_normal((500, 500))
    C = gpu.dgemm(A, R)
    ref = A @ R
    assert np.max(np.abs(C - ref)) <= 1e-11 * np.max(np.abs(ref))
    del A, C, ref
    W = rng.standard_normal((1001, 250000))
    G = gpu.dgemm(W, W, layB=0, lower_only=True)
    ref = W @ W.T
    il = np.tril_indices(1001)
    assert np.max(np.abs(G[il] - ref[il])) <= 1e-11 * np.max(np.abs(ref))


@pytest.mark.parametrize("n", [2, 5, 16, 17, 24, 25, 33, 43, 48, 63, 64])
def test_small_block_step_length_eigenvalue(gpu, n):
    """lambda_min(L D L^T) by the small-block kernels through the multi-block launch the engine uses (n <= 16: exact in one wavefront;
    17 .. 48: exact by Householder reduction + multisection, the half width of the last interval as residual; above: single-launch
    Lanczos, Ritz value and its residual bound): theta - resid <= lambda_min <= theta + resid, both slots of the launch and both
    blocks of the job table agree"""
    import ctypes as C
    lib = gpu.ulib()
    rng = np.random.default_rng(n)
    for trial in range(5):
        L = np.tril(rng.standard_normal((n, n))) + 2.0 * np.eye(n)
        G = rng.standard_normal((n, n))
        D = (G + G.T) / 2 - (0.5 * trial) * np.eye(n)
        if trial >= 3:
            # the regime of an interior-point iteration: L = inverse Cholesky factor of a nearly singular X, eigenvalues of
            # L D L^T spread over ten orders of magnitude (|alpha_j| >> beta_j in the Lanczos recurrence)
            L = np.diag(10.0 ** np.linspace(0, 5, n)) @ L
        ref = float(np.linalg.eigvalsh(L @ D @ L.T)[0])
        th = np.zeros(2)
        rs = np.zeros(2)
        Lc, Dc = np.ascontiguousarray(L), np.ascontiguousarray(D)
        rc = lib.hipsdp_lambda_min_scaled(0, n, Lc.ctypes.data_as(C.POINTER(C.c_double)), Dc.ctypes.data_as(C.POINTER(C.c_double)), 24,
                                          th.ctypes.data_as(C.POINTER(C.c_double)), rs.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0
        assert th[0] == th[1] and rs[0] == rs[1]
        scale = np.abs(np.linalg.eigvalsh(L @ D @ L.T)).max()
        assert th[0] + rs[0] >= ref - 1e-9 * scale, (th, rs, ref)
        assert th[0] - rs[0] <= ref + 1e-9 * scale, (th, rs, ref)
        if n <= 48:
            assert abs(th[0] - ref) <= 1e-8 * scale          # n <= 16 and 17 .. 48: the eigenvalue itself (reduction + multisection)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 16, 17, 33, 50, 64, 65, 66, 100, 127, 128])
def test_ith_eigenpair_in_one_launch_matches_dsyevr_range_i(gpu, n):
    """SCIPlapackComputeIthEigenvalue at the sizes its callers use (cons_sdp.c: blocks of 2-50 rows; lapack_interface.c:178-288):
    n <= 64 runs k_syevi_small (Householder tridiagonalisation, Sturm multisection for exactly the i-th eigenvalue, inverse
    iteration + back-transformation) through pinned staging memory, 64 < n <= 128 k_syevi_mid (the same with the matrix in LDS).  Every i: eigenvalue against numpy, eigenvector by residual and
    norm (sign and the basis of a multiple eigenvalue are free, as with DSYEVR)."""
    lib = gpu.lib()
    rng = np.random.default_rng(40 + n)
    G = rng.standard_normal((n, n))
    A = 0.5 * (G + G.T)
    if n >= 5:
        A[2, :] = A[:, 2] = 0.0
        A[2, 2] = A[3, 3]                      # stress: a decoupled row and (nearly) repeated diagonal entries
    ev = np.linalg.eigvalsh(A)
    scale = max(1.0, np.abs(ev).max())
    for i in range(1, n + 1):
        val = C.c_double(0.0)
        vec = np.zeros(n)
        assert lib.SCIPlapackComputeIthEigenvalue(None, 1, n, _pd(A.copy().reshape(-1)), i, C.byref(val), _pd(vec)) == 1
        assert abs(val.value - ev[i - 1]) <= 1e-12 * scale * n
        assert abs(np.linalg.norm(vec) - 1.0) <= 1e-12
        assert np.linalg.norm(A @ vec - val.value * vec) <= 1e-9 * scale
        val2 = C.c_double(0.0)
        assert lib.SCIPlapackComputeIthEigenvalue(None, 0, n, _pd(A.copy().reshape(-1)), i, C.byref(val2), None) == 1
        assert val2.value == val.value
    # only the triangle DSYEVR('L') reads from a column-major array (memory [j n + i], i >= j) is used
    B = np.triu(A) + np.tril(rng.standard_normal((n, n)), -1)
    val = C.c_double(0.0)
    assert lib.SCIPlapackComputeIthEigenvalue(None, 0, n, _pd(B.reshape(-1).copy()), 1, C.byref(val), None) == 1
    assert abs(val.value - ev[0]) <= 1e-12 * scale * n
    # repeated eigenvalues: 2 I (+) 5 I
    D = np.diag([2.0] * (n // 2) + [5.0] * (n - n // 2))
    for i in range(1, n + 1):
        vec = np.zeros(n)
        assert lib.SCIPlapackComputeIthEigenvalue(None, 1, n, _pd(D.copy().reshape(-1)), i, C.byref(val), _pd(vec)) == 1
        assert abs(val.value - np.diag(D)[i - 1]) <= 1e-13 and np.linalg.norm(D @ vec - val.value * vec) <= 1e-10


def test_small_eigenvalue_calls_stay_out_of_the_millisecond_regime(gpu):
    """the path the callers hammer (dozens of calls per node): no allocation, no copy engine, no stream synchronisation - a call
    at n = 30 must not cost what a hipMalloc + hipMemcpy + full decomposition costs (milliseconds before); the measured numbers
    are in DESIGN.md (tests/devtools/lapack_small_time.py)"""
    import time
    lib = gpu.lib()
    rng = np.random.default_rng(1)
    G = rng.standard_normal((30, 30))
    A = (G + G.T).reshape(-1).copy()
    val = C.c_double(0.0)
    for _ in range(5):
        lib.SCIPlapackComputeIthEigenvalue(None, 0, 30, _pd(A), 1, C.byref(val), None)
    t0 = time.perf_counter()
    for _ in range(200):
        lib.SCIPlapackComputeIthEigenvalue(None, 0, 30, _pd(A), 1, C.byref(val), None)
    per = (time.perf_counter() - t0) / 200
    assert per < 5e-4, per


@pytest.mark.parametrize("R,E", [(501, 2500), (1001, 5050), (301, 16900), (2001, 10000), (129, 9000), (64, 400), (1001, 70000)])
def test_pass_AT_in_row_chunks_matches_the_plain_kernel(gpu, R, E):
    """hs_gemv_t_ws: blocks with few entries and many rows are summed in row chunks side by side (two launches) instead of one thread
    per pair of entries walking all rows.  Against numpy and against the plain kernel (another summation order: 1e-13 relative), and
    twice with the same bits; the last two shapes do not split (too little work / too many entries)."""
    lib = gpu.ulib()
    rng = np.random.default_rng(R + E)
    A = rng.standard_normal((R, E))
    coef = rng.standard_normal(R)
    add = rng.standard_normal(E)
    ref = coef @ A + 0.75 * add
    outs = []
    chunks = C.c_int(-1)
    for split in (0, 1, 1):
        out = np.zeros(E)
        lib.hipsdp_pass_at_unit.argtypes = [C.c_int, C.c_int, C.c_longlong, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                      C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        assert lib.hipsdp_pass_at_unit(0, R, E, _pd(A.reshape(-1)), _pd(coef), 0.75, _pd(add), split, _pd(out), C.byref(chunks)) == 0
        outs.append(out)
    scale = np.abs(A).T @ np.abs(coef) + np.abs(add)
    for out in outs:
        assert np.max(np.abs(out - ref) / scale) <= 1e-14 * np.sqrt(R) + 1e-15
    assert np.array_equal(outs[1], outs[2])
    if (R, E) in ((64, 400), (1001, 70000)):
        assert chunks.value == 0 and np.array_equal(outs[0], outs[1])
    else:
        assert chunks.value >= 2

