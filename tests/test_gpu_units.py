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


@pytest.mark.parametrize("n", [1, 2, 7, 63, 64, 65, 130, 300, 777])
def test_cholesky_inverse_and_solves(gpu, n):
    G = RNG.standard_normal((n, n))
    S = G @ G.T + n * np.eye(n)
    L, fail = gpu.potrf(S)
    assert fail == 0
    assert rel(L, np.linalg.cholesky(S)) <= 1e-13
    assert rel(gpu.trtri(S), np.linalg.inv(np.linalg.cholesky(S))) <= 1e-12
    r = RNG.standard_normal((3, n))
    assert rel(gpu.potrs(S, r), np.linalg.solve(S, r.T).T) <= 1e-11


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


@pytest.mark.parametrize("n", [1, 2, 3, 10, 43, 128])
def test_syev_matches_dsyevr_semantics(gpu, n):
    """ascending eigenvalues, eigenvectors as ROWS (lapack_interface.c:507-603)"""
    G = RNG.standard_normal((n, n))
    W = G + G.T
    lam, V = gpu.syev(W)
    ev = np.linalg.eigvalsh(W)
    assert rel(lam, ev) <= 1e-12
    assert np.abs(V @ V.T - np.eye(n)).max() <= 1e-12
    assert np.abs(V @ W @ V.T - np.diag(lam)).max() <= 1e-11 * max(1.0, np.abs(ev).max())


@pytest.mark.parametrize("R,E", [(1, 1), (5, 100), (37, 10000), (300, 40001), (1001, 2500)])
def test_gemv_passes(gpu, R, E):
    A = RNG.standard_normal((R, E))
    V = RNG.standard_normal((3, E))
    c = RNG.standard_normal(R)
    assert rel(gpu.gemv_n(A, V), V @ A.T) <= 1e-13 * E ** 0.5
    assert rel(gpu.gemv_t(A, c), c @ A) <= 1e-13 * R ** 0.5
