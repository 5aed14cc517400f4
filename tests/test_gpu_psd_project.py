"""Row a22 / (f)-3 of SURVEY.md section 8: the PSD projection chain of the warm-start producer (relax_sdp.c:2715-2766 for Z,
:3405-3445 for X).  Three ways to the same numbers:
  (1) the reference's chain composed from the drop-in primitives, exactly as relax_sdp.c composes it:
      SCIPlapackComputeEigenvectorDecomposition -> clamp -> scaleTransposedMatrix -> SCIPlapackMatrixMatrixMult(V, TRUE, S, FALSE);
  (2) oracle/psd_project_ref.chain (numpy restatement of that chain) fed with the SAME eigenvectors - the literal chain weights
      the components of the eigenvectors, so its result depends on the basis the eigen-solver returns and can only be compared on
      a common basis;
  (3) hipsdp_psd_project, the device-fused form (one upload, one download), mode 0 = literal chain, mode 1 = spectral form
      (basis independent, compared with numpy's own decomposition)."""
import ctypes as C
import numpy as np
import pytest

import psd_project_ref as ref

pytestmark = pytest.mark.gpu
PD = C.POINTER(C.c_double)


def _pd(a):
    return a.ctypes.data_as(PD)


def random_sparse_sym(n, seed, density):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    M = 0.5 * (M + M.T)
    mask = rng.random((n, n)) < density
    mask = np.triu(mask) | np.triu(mask).T | np.eye(n, dtype=bool)
    M = M * mask
    r, c = np.nonzero(np.tril(M))
    return r.astype(np.int32), c.astype(np.int32), M[r, c].copy(), M


def device_eig(lib, n):
    def eig(full_flat):
        a = np.array(full_flat, dtype=np.float64).copy()
        lam = np.zeros(n)
        V = np.zeros(n * n)
        assert lib.SCIPlapackComputeEigenvectorDecomposition(None, n, _pd(a), _pd(lam), _pd(V)) == 1
        return lam, V
    return eig


def chain_through_primitives(lib, n, row, col, val, minev, eps=ref.EPSILON):
    """relax_sdp.c:2733-2766 with the SCIPlapack* entry points of libhipsdp.so"""
    full = ref.expand_sparse(n, row, col, val)
    lam = np.zeros(n)
    V = np.zeros(n * n)
    assert lib.SCIPlapackComputeEigenvectorDecomposition(None, n, _pd(full.copy()), _pd(lam), _pd(V)) == 1
    scaled = V.copy()
    i = 0
    while i < n and lam[i] - minev < -eps:
        lam[i] = minev
        i += 1
    scaled = (scaled.reshape(n, n) * lam[None, :]).reshape(-1).copy()      # scaleTransposedMatrix: entry [r][c] *= scale[c]
    out = np.zeros(n * n)
    assert lib.SCIPlapackMatrixMatrixMult(n, n, _pd(V), 1, n, n, _pd(scaled), 0, _pd(out)) == 1
    return out.reshape(n, n)


@pytest.mark.parametrize("n,density", [(10, 1.0), (43, 0.3), (128, 0.1), (500, 0.05)])
def test_psd_projection_chain(gpu, n, density):
    lib = gpu.lib()
    row, col, val, M = random_sparse_sym(n, 100 + n, density)
    minev = 1e-4
    scale = max(1.0, np.abs(M).max() * n ** 0.5)
    # (1) vs (2): the chain through the drop-in primitives equals the restated chain on the device's own eigenvectors
    R1 = chain_through_primitives(lib, n, row, col, val, minev)
    r2, c2, v2, R2 = ref.chain(n, row, col, val, minev, eig=device_eig(lib, n))
    assert np.max(np.abs(R1 - R2)) <= 1e-10 * scale
    # (3) fused, literal mode: same entries, same order
    r3, c3, v3 = gpu.psd_project(n, row, col, val, minev, ref.EPSILON, 0)
    D3 = np.zeros((n, n))
    D3[r3, c3] = v3
    D2 = np.zeros((n, n))
    D2[r2, c2] = v2
    assert np.max(np.abs(D3 - D2)) <= 1e-10 * scale
    assert np.all(r3 <= c3) and np.all(np.diff(r3.astype(np.int64) * n + c3) > 0)       # upper triangle, row-major order
    # the literal chain still yields a matrix whose eigenvalues are the clamped ones (V is orthogonal either way)
    lam = np.linalg.eigvalsh(M)
    lam_c = np.sort(np.where(lam - minev < -ref.EPSILON, minev, lam))
    full3 = D3 + np.triu(D3, 1).T
    assert np.max(np.abs(np.linalg.eigvalsh(full3) - lam_c)) <= 1e-8 * scale
    # (3) fused, spectral mode vs numpy (basis independent): the projection onto {X : X >= minev I}
    r4, c4, v4 = gpu.psd_project(n, row, col, val, minev, ref.EPSILON, 1)
    D4 = np.zeros((n, n))
    D4[r4, c4] = v4
    S = ref.spectral(n, row, col, val, minev)
    keep = np.abs(np.triu(S)) > ref.EPSILON
    assert np.max(np.abs((D4 - np.triu(S))[keep])) <= 1e-9 * scale
    assert np.linalg.eigvalsh(D4 + np.triu(D4, 1).T).min() >= minev - 1e-8 * scale


def test_psd_projection_edge_cases(gpu):
    # an already PSD diagonal matrix comes back unchanged; an empty matrix becomes minev * I; a too small buffer reports the need
    n = 6
    d = np.arange(1, n + 1, dtype=np.float64)
    idx = np.arange(n, dtype=np.int32)
    r, c, v = gpu.psd_project(n, idx, idx, d, 1e-3, 1e-9, 0)
    assert list(r) == list(range(n)) and list(c) == list(range(n)) and np.allclose(v, d, atol=1e-12)
    r, c, v = gpu.psd_project(n, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), 0.5, 1e-9, 1)
    assert list(r) == list(range(n)) and np.allclose(v, 0.5)
    lib = gpu.lib()
    k = C.c_int(0)
    ro = np.zeros(2, dtype=np.int32)
    vo = np.zeros(2)
    rc = lib.hipsdp_psd_project(0, n, n, idx.ctypes.data_as(C.POINTER(C.c_int)), idx.ctypes.data_as(C.POINTER(C.c_int)), _pd(d),
                                C.c_double(1e-3), C.c_double(1e-9), 0, 2, C.byref(k), ro.ctypes.data_as(C.POINTER(C.c_int)),
                                ro.ctypes.data_as(C.POINTER(C.c_int)), _pd(vo))
    assert rc != 0 and k.value == n
