"""rank-deficient Schur-like matrices: the engine's blocked semidefinite Cholesky (hipsdp_potrf_ex, psd mode) against the oracle's
chol_psd - forced pivots, reconstruction L L^T and the solution of a consistent system (developer tool, GPU box)"""
import importlib.util, os, sys
import numpy as np
import scipy.linalg as sla
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref

rng = np.random.default_rng(3)
for m, rank, spread in ((200, 9, 0), (200, 9, 8), (129, 100, 6), (129, 129, 12), (260, 40, 10)):
    B = rng.standard_normal((m, rank)) * (10.0 ** (-spread * rng.random(rank)))[None, :]
    M = B @ B.T
    M = 0.5 * (M + M.T)
    Lo = ipm_ref.chol_psd(M)
    Lg, dinv, mask, fail = hb.potrf_ex(M, psd=True)
    Lg = np.tril(Lg)
    fo = np.flatnonzero(np.diag(Lo) ** 2 <= 1.0000001e-13 * np.diag(M))
    fg = np.flatnonzero(np.diag(Lg) ** 2 <= 1.0000001e-13 * np.diag(M))
    zo = sum(1 for k in fo if not np.any(Lo[k + 1:, k]))
    zg = int(np.sum(mask))
    print("   forced oracle %d (zeroed %d), engine %d (zeroed %d); first forced columns oracle %s engine %s" % (len(fo), zo, len(fg), zg, fo[:8], fg[:8]))
    b = M @ rng.standard_normal(m)
    xo = sla.solve_triangular(Lo.T, sla.solve_triangular(Lo, b, lower=True), lower=False)
    xg = sla.solve_triangular(Lg.T, sla.solve_triangular(Lg, b, lower=True), lower=False)
    nM = np.linalg.norm(M)
    print("m %d rank %d spread 1e-%d: forced pivots oracle %d engine(mask) %d | |LL^T - M|/|M| oracle %.1e engine %.1e | residual |Mx-b|/|b| oracle %.1e engine %.1e | |x| oracle %.2e engine %.2e"
          % (m, rank, spread, len(fo), len(fg), np.linalg.norm(Lo @ Lo.T - M) / nM, np.linalg.norm(Lg @ Lg.T - M) / nM,
             np.linalg.norm(M @ xo - b) / np.linalg.norm(b), np.linalg.norm(M @ xg - b) / np.linalg.norm(b), np.linalg.norm(xo), np.linalg.norm(xg)))
