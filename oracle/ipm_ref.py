"""ipm_ref.py - TEST INFRASTRUCTURE (oracle).  CPU/numpy restatement of the interior-point method that the HIP engine
(scip-sdp_amd/csrc/ipm.hip) runs on the GPU.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file; the product path never does.

Parity status: the arithmetic of the reference's node solve lives in DSDP 5.8 / SDPA 7.4.4 / MOSEK (INSTALL:10-12), none
of which is vendored under /root/reference (the reference only marshals into them: src/sdpi/sdpisolver_dsdp.c:1077-1130,
1503; src/sdpi/sdpisolver_sdpa.cpp:1179-1273, 1620).  The iterates are therefore "parity unpinned"; what IS pinned is the
result at the SDPI boundary: the known answers of unittests/src/checksdpi.c (tests 1-4, 9-11), the optima of
check/testset/short.solu, and the acceptance conditions of src/sdpi/sdpsolchecker.c:58-265 (restated in checker.py).
tests/test_oracle_golden.py checks this file against all of them.

Problem form (src/sdpi/sdpisolver.h:37-42, src/sdpi/sdpi.c:39-58), after the backend's marshalling:

   (D)  min  b^T y   s.t.  Z_k = sum_i A_i^k y_i - A_0^k  psd  (k = 1..K),   z = D y - c >= 0
   (P)  max  sum_k <A_0^k, X_k> + c^T x   s.t.  sum_k <A_i^k, X_k> + (D^T x)_i = b_i,  X_k psd,  x >= 0

every finite side of an LP row and every finite variable bound is one row of (D, c).

Algorithm: homogeneous self-dual embedding (tau, kappa) with the HKM direction, Mehrotra predictor-corrector; the Schur
complement is  M_ij = sum_k tr(A_i^k X_k A_j^k Z_k^-1) + (D^T diag(x/z) D)_ij  computed for i, j = 0..m ("variable 0" is
the constant matrix), which yields M, g = M[0,1:] and omega = M[0,0] in one assembly.
"""
import numpy as np
import scipy.linalg as sla


class CoreProblem:
    """Marshalled problem: b[m]; blocks: list of arrays A[m+1, n, n] (A[0] = constant matrix A_0, A[i] = A_i);
    D[q, m], c[q] (rows D y - c >= 0)."""

    def __init__(self, b, blocks, D=None, c=None):
        self.b = np.asarray(b, dtype=np.float64)
        self.m = self.b.shape[0]
        self.blocks = [np.ascontiguousarray(A, dtype=np.float64) for A in blocks]
        for A in self.blocks:
            assert A.shape[0] == self.m + 1 and A.shape[1] == A.shape[2]
        if D is None:
            D = np.zeros((0, self.m))
            c = np.zeros(0)
        self.D = np.ascontiguousarray(D, dtype=np.float64).reshape(-1, self.m)
        self.c = np.asarray(c, dtype=np.float64).reshape(-1)
        self.q = self.D.shape[0]
        assert self.c.shape[0] == self.q

    @property
    def N(self):
        return sum(A.shape[1] for A in self.blocks) + self.q


class Params:
    def __init__(self, gaptol=1e-5, feastol=1e-5, maxiter=100, gamma=0.98, verbose=False, infeastol=1e-7, pabstol=0.0, preoptgap=0.0,
                 settings=0):
        # retry ladder of the backend (SCIP_SDPSOLVERSETTING, type_sdpi.h:69-77; sdpisolver_sdpa.cpp:1415-1449, 1698-1795):
        # 0 fast, 1 medium, 2 stable - shorter steps, a higher centrality floor, more patient stall tests (same numbers as
        # csrc/ipm.hip: solve_impl)
        self.settings = settings
        # 'U': Mx = <A_i, X A_j Z^-1> by three full products (default); 'W': the device path's formulation with the triangular
        # savings (schur_block_w) - the CPU baseline of bench.py
        self.schur = 'U'
        self.preoptgap = preoptgap    # > 0: keep the first feasible iterate with relative gap below it (Result.pre)
        self.gaptol = gaptol          # relative gap / absolute gap tolerance (relax_sdp.c:70)
        self.feastol = feastol        # residual tolerance (relax_sdp.c:71)
        self.maxiter = maxiter
        self.gamma = gamma            # fraction of the step to the boundary
        self.verbose = verbose
        self.infeastol = infeastol    # tolerance of the Farkas certificates
        self.pabstol = pabstol        # > 0: optimal termination also needs ||b - A(X)||_2 <= pabstol (absolute, the bound
                                      # sdpsolchecker.c:775-931 applies with SCIP_SDPPAR_FEASTOL)


class Result:
    pass


STATUS_OPTIMAL = 0        # both problems feasible, converged
STATUS_DINF = 1           # (D) infeasible: X-ray with <A_0,X> + c^T x > 0, A(X,x) = 0
STATUS_DUNB = 2           # (D) unbounded / (P) infeasible: y-ray with b^T y < 0, A^T y psd
STATUS_PDINF = 3          # both certificates
STATUS_ITERLIM = 4
STATUS_NUMERIC = 5


def sym(M):
    return 0.5 * (M + M.T)


def schur_block(A, X, Zinv):
    """Extended Schur contribution of one dense block: Mx[i, j] = tr(A_i X A_j Zinv), i, j = 0..m.

    Three large BLAS calls, like the device path (T_j = A_j Zinv, U_j = X T_j, Mx = A_flat U_flat^T).  The batched
    left-multiplication is written as one stack product: V_j = T_j^T X = (X T_j)^T, and because every A_i is symmetric
    <A_i, U_j> = <A_i, U_j^T>, so Mx = A_flat V_flat^T needs no explicit transposition of the result."""
    m1, n, _ = A.shape
    T = A.reshape(m1 * n, n) @ Zinv                                        # GEMM1: stack of A_j times Zinv
    Tt = np.ascontiguousarray(T.reshape(m1, n, n).transpose(0, 2, 1))      # T_j^T (one memory pass)
    V = Tt.reshape(m1 * n, n) @ X                                          # GEMM2: stack product, V_j = T_j^T X
    Mx = A.reshape(m1, n * n) @ V.reshape(m1, n * n).T                     # GEMM3
    return 0.5 * (Mx + Mx.T)


def schur_rows_sparse(m, n, coo, X, Zinv):
    """Schur entries tr(A_i X A_j Zinv), i, j = 1 .. m, from the nonzeros in the association of the dense formula - what
    scip-sdp_amd/csrc/sparse.hip (k_sp_trows, k_sp_schur) and csrc/solve1.hip compute since round 4: T_j = A_j Zinv restricted to the
    non-empty rows of A_j, then M_ij = sum over the entries (a, b) of A_i (both triangles) of A_i[a][b] sum_p X[b][p] T_j[p][a].
    coo: lower-triangular triplets (var 1 .. m, row >= col, val).  Plain loops (small cases only).  Returns the m x m matrix."""
    var, row, col, val = coo
    full = [[] for _ in range(m + 1)]
    for v, r, c, x in zip(var, row, col, val):
        full[int(v)].append((int(r), int(c), float(x)))
        if r != c:
            full[int(v)].append((int(c), int(r), float(x)))
    T = []
    for j in range(1, m + 1):
        rows = {}
        for (p_, q_, x) in sorted(full[j]):
            rows.setdefault(p_, np.zeros(n))
            rows[p_] = rows[p_] + x * Zinv[q_, :]
        T.append(sorted(rows.items()))
    M = np.zeros((m, m))
    for i in range(1, m + 1):
        ents = sorted(full[i])
        for j in range(1, m + 1):
            acc = 0.0
            for (a, b_, x) in ents:
                u = 0.0
                for (p_, trow) in T[j - 1]:
                    u += X[b_, p_] * trow[a]
                acc += x * u
            M[i - 1, j - 1] = acc
    return M


def schur_pairs_sparse(m, coo, X, Zinv):
    """Schur entries tr(A_i X A_j Zinv), i, j = 1 .. m, from the nonzeros alone by the PAIR formula (SDPA's F3 case; what
    scip-sdp_amd/csrc/sparse.hip used in rounds 2-3 - the same numbers in exact arithmetic, but see schur_rows_sparse and DESIGN.md
    7.3): with e = (p, q, a) in A_i and f = (r, s, b) in A_j (lower triangles, off-diagonal entries stand for both
    positions)   sum_e sum_f a b (X_qr Zinv_sp + [p != q] X_pr Zinv_sq + [r != s] X_qs Zinv_rp + [p != q][r != s] X_ps Zinv_rq).
    Vectorised over all pairs of entries (fine for the sizes the CPU tests use).  Returns the m x m matrix."""
    var, row, col, val = coo
    p, q = row[:, None], col[:, None]
    r, s = row[None, :], col[None, :]
    w = X[q, r] * Zinv[s, p]
    w = w + np.where(p != q, X[p, r] * Zinv[s, q], 0.0)
    w = w + np.where(r != s, X[q, s] * Zinv[r, p], 0.0)
    w = w + np.where((p != q) & (r != s), X[p, s] * Zinv[r, q], 0.0)
    w = w * (val[:, None] * val[None, :])
    M = np.zeros((m, m))
    np.add.at(M, (var[:, None] - 1 + 0 * var[None, :], 0 * var[:, None] + var[None, :] - 1), w)
    return M


def schur_block_w(A, Lx, Lz):
    """The same matrix through the W formulation of the device path (csrc/schur.hip: hs_schur_W) with the triangular savings kept,
    for the CPU baseline of bench.py: X = R R^T (R = chol X), Z^-1 = G^T G (G = inverse of chol Z), W_j = G A_j R,
    Mx = W W^T over the n^2 entries.  Level-3 BLAS only: two DTRMM over stacks (n^3 m1 each instead of 2 n^3 m1) and one DSYRK
    (m1^2 n^2 instead of 2 m1^2 n^2).  The row-major stacks are handed to the Fortran routines as their transposes (no copies
    besides the two the products overwrite):  T^T = R^T A_stack^T,  (W_j^T)_stack^T = G (T_j^T)_stack^T,  <W_i, W_j> = <W_i^T, W_j^T>."""
    m1, n, _ = A.shape
    G = sla.solve_triangular(Lz, np.eye(n), lower=True, check_finite=False)          # lower triangular
    # T = A_stack R:  as Fortran arrays  T^T (n x m1 n) = R^T (upper) * A_stack^T
    T = np.array(A.reshape(m1 * n, n), order='C', copy=True)
    Tt_f = sla.blas.dtrmm(1.0, Lx.T, T.T, side=0, lower=0, trans_a=0, overwrite_b=1)   # Lx.T is the F-view of a C-ordered lower R: upper
    T = Tt_f.T.reshape(m1, n, n)
    # W_j^T = T_j^T G^T: stack of T_j^T (one memory pass), times G^T from the right:  (stack)^T (n x m1 n) = G (lower) * (T_j^T stack)^T
    Tt = np.ascontiguousarray(T.transpose(0, 2, 1)).reshape(m1 * n, n)
    Wt_f = sla.blas.dtrmm(1.0, np.asfortranarray(G), Tt.T, side=0, lower=1, trans_a=0, overwrite_b=1)
    Wf = Wt_f.T.reshape(m1, n * n)                                                  # row j = vec(W_j^T)
    # Mx = Wf Wf^T: DSYRK on the F-view a = Wf^T (n^2 x m1): a^T a
    Mx = sla.blas.dsyrk(1.0, Wf.T, trans=1, lower=1)
    Mx = np.tril(Mx)
    return Mx + np.tril(Mx, -1).T


import os as _os
# 0: keep forced columns, 1: zero all forced columns, 2: zero those with a non-positive pivot, 3 (default): zero those whose pivot is at
# rounding-noise level, <= 8 eps (k + 1) M_kk - the sign of a noise pivot differs between two implementations of the same algorithm
# (this file, scip-sdp_amd/csrc/chol.hip), its size does not: 400-problem stress family with the settings ladder, engine against
# oracle: 3 status differences with rule 2, none with rule 3 (profiles/r03_b_stress_pivot_rules.txt)
PIVOT_RULE = int(_os.environ.get('HIPSDP_PIVOT_RULE', '3'))


def chol_psd(M, regtol=1e-13):
    """Cholesky of a positive SEMI-definite matrix (the Schur complement has dependent columns when the constraint
    matrices are linearly dependent): a pivot below regtol * M_kk is replaced by regtol * M_kk (a small positive pivot keeps
    the direction alive, so that an unbounded ray along it can still be found).  When the pivot is not even positive the
    rest of that column is rounding noise of a column that is zero in exact arithmetic and is set to zero: dividing the noise
    by the forced pivot and eliminating with it amplifies it exponentially over a run of dependent columns (600 random
    problems of tests/devtools/stress_gpu.py: 27 numerical failures without, 6 with).  Same rule as k_potrf_diag
    (scip-sdp_amd/csrc/chol.hip) in semidefinite mode."""
    try:
        L = np.linalg.cholesky(M)
        if np.all(np.diag(L) ** 2 > regtol * np.diag(M)):
            return L
    except np.linalg.LinAlgError:
        pass
    n = M.shape[0]
    L = np.tril(M).astype(np.float64).copy()
    for k in range(n):
        d = L[k, k]
        if not (d > regtol * M[k, k]) or not (d > 1e-300):
            zero = (PIVOT_RULE == 1 or (PIVOT_RULE == 2 and not (d > 0.0))
                    or (PIVOT_RULE == 3 and not (d > 1.78e-15 * (k + 1) * M[k, k])))
            d = regtol * M[k, k] if M[k, k] > 1e-280 else 1.0
            if zero:
                L[k, k] = np.sqrt(d)
                L[k + 1:, k] = 0.0
                continue
        L[k, k] = np.sqrt(d)
        L[k + 1:, k] /= L[k, k]
        L[k + 1:, k + 1:] -= np.tril(np.outer(L[k + 1:, k], L[k + 1:, k]))
    return L


def max_step_psd(L, dX):
    """largest alpha with  L L^T + alpha dX  psd  (inf if dX psd):  -1 / lambda_min(L^-1 dX L^-T)."""
    W = sla.solve_triangular(L, dX, lower=True, check_finite=False)
    W = sla.solve_triangular(L, W.T, lower=True, check_finite=False)
    if not np.all(np.isfinite(W)):
        return np.nan
    lam = sla.eigh(sym(W), eigvals_only=True, subset_by_index=[0, 0], check_finite=False)[0]
    return np.inf if lam >= 0 else -1.0 / lam


def max_step_vec(x, dx):
    neg = dx < 0
    if not np.any(neg):
        return np.inf
    return float(np.min(-x[neg] / dx[neg]))


def warm_start_point(prob, y, X, Z, x, z):
    """start tuple for hsd_solve from a caller-supplied point (sdpisolver.h:160-173; consumed as in
    sdpisolver_sdpa.cpp:1481-1592) or None: the point is used only if it is strictly interior (X_k, Z_k positive definite,
    x, z > 0); tau = 1, kappa = mean complementarity"""
    X = [sym(np.asarray(Xk, dtype=np.float64)) for Xk in X]
    Z = [sym(np.asarray(Zk, dtype=np.float64)) for Zk in Z]
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    z = np.asarray(z, dtype=np.float64).reshape(-1)
    try:
        for M in X + Z:
            np.linalg.cholesky(M)
    except np.linalg.LinAlgError:
        return None
    if prob.q and (x.min() <= 0.0 or z.min() <= 0.0):
        return None
    N = prob.N
    mu0 = (sum(np.sum(Xk * Zk) for Xk, Zk in zip(X, Z)) + (x @ z if prob.q else 0.0)) / max(N, 1)
    if not np.isfinite(mu0) or mu0 <= 0.0:
        return None
    return (np.asarray(y, dtype=np.float64).copy(), X, Z, x.copy(), z.copy(), 1.0, float(mu0))


def hsd_solve(prob, par=None, start=None):
    """Solves the core problem.  Returns Result with y, X (list), Z (list), x, z scaled back by tau, status, iterations."""
    par = par or Params()
    m, q, K = prob.m, prob.q, len(prob.blocks)
    b, D, c = prob.b, prob.D, prob.c
    Dext = np.concatenate([c.reshape(-1, 1), D], axis=1)      # [q, m+1], column 0 = constant
    ns = [A.shape[1] for A in prob.blocks]
    Aflat = [A.reshape(m + 1, -1) for A in prob.blocks]
    N1 = prob.N + 1

    normb = np.linalg.norm(b)
    normC = np.sqrt(sum(np.sum(A[0] ** 2) for A in prob.blocks) + np.sum(c ** 2))

    # starting point
    if start is None:
        xi = max(1.0, np.sqrt(max(normb, normC, 1.0)))
        y = np.zeros(m)
        X = [xi * np.eye(n) for n in ns]
        Z = [xi * np.eye(n) for n in ns]
        x = xi * np.ones(q)
        z = xi * np.ones(q)
        tau, kappa = 1.0, xi * xi
    else:
        y, X, Z, x, z, tau, kappa = start

    settings = min(2, max(0, int(par.settings)))
    gamma_eff = par.gamma if settings == 0 else min(par.gamma, 0.9 if settings == 1 else 0.75)
    stall_lim = (3, 5, 8)[settings]
    nobest_lim = (6, 10, 15)[settings]
    sigma_floor = (1e-8, 1e-4, 1e-2)[settings]
    maxiter = par.maxiter

    res = Result()
    res.settings_used = settings
    res.status = STATUS_ITERLIM
    res.history = []
    res.pre = None
    it = 0
    certwait = 0
    nstall = 0
    lastmu = np.inf
    alpha_last = 1.0
    bestmerit = np.inf
    sincebest = 0
    for it in range(maxiter + 1):
        # ---- residuals
        AX = sum(Af @ Xk.reshape(-1) for Af, Xk in zip(Aflat, X)) if K else np.zeros(m + 1)
        AX = AX + Dext.T @ x                                     # [<A_0,X> + c^T x ; A(X,x)]
        rp = b * tau - AX[1:]
        yt = np.concatenate([[-tau], y])
        Rd = [np.tensordot(yt, A, axes=(0, 0)) - Zk for A, Zk in zip(prob.blocks, Z)]
        rd = Dext @ yt - z
        pobj = AX[0]                       # <A_0,X> + c^T x   (times tau scaling below)
        dobj = b @ y
        rg = pobj - dobj - kappa
        mu = (sum(np.sum(Xk * Zk) for Xk, Zk in zip(X, Z)) + x @ z + tau * kappa) / N1

        pinf = np.linalg.norm(rp) / tau / (1.0 + normb)
        dinf = np.sqrt(sum(np.sum(R ** 2) for R in Rd) + np.sum(rd ** 2)) / tau / (1.0 + normC)
        # absolute violation of (D) by y / tau, the quantity sdpsolchecker.c:201-257 bounds (|lambda_min| <= ||R_d||_F)
        dabs = max([np.sqrt(np.sum(R ** 2)) for R in Rd] + [np.max(np.abs(rd)) if q else 0.0]) / tau
        gap = abs(dobj - pobj) / tau
        res.history.append((it, mu, pinf, dinf, gap, tau, kappa))
        if par.verbose:
            print("it %3d mu %.3e pinf %.3e dinf %.3e gap %.3e pobj %.8e dobj %.8e tau %.3e kap %.3e" %
                  (it, mu, pinf, dinf, gap, pobj / tau, dobj / tau, tau, kappa))

        # ---- preoptimal iterate: first one feasible to tolerance with relative gap below preoptgap (capture rule of
        # sdpisolver_dsdp.c:323-358; SCIP_SDPPAR_WARMSTARTPOGAP)
        if par.preoptgap > 0.0 and res.pre is None and pinf <= par.feastol and dabs <= par.feastol \
                and gap / (1.0 + 0.5 * abs(pobj / tau) + 0.5 * abs(dobj / tau)) < par.preoptgap:
            res.pre = dict(it=it, y=y / tau, X=[Xk / tau for Xk in X], x=x / tau)

        # ---- termination
        # optimal: absolute gap (sdpisolver_dsdp.c:1558-1571) and feasibility of y / tau within feastol
        pabs = np.linalg.norm(rp) / tau
        pabsok = par.pabstol <= 0.0 or pabs <= par.pabstol
        if pinf <= par.feastol and pabsok and dabs <= par.feastol and gap <= par.gaptol:
            res.status = STATUS_OPTIMAL
            break
        # Farkas certificates, scale free: homogeneous residual relative to the objective value it certifies
        if tau < 1e-2 * min(1.0, kappa) or mu / (tau * tau) > 1e10:
            hd = np.sqrt(sum(np.sum((R + tau * A[0]) ** 2) for R, A in zip(Rd, prob.blocks))
                         + np.sum((rd + tau * c) ** 2))              # || A^T y - Z ||
            hp = np.linalg.norm(AX[1:])                                # || A(X,x) ||
            big = max(abs(dobj), abs(pobj))
            cand_dunb = dobj < -1e-3 * big                              # y-ray candidate:  b^T y < 0
            cand_dinf = pobj > 1e-3 * big                               # X-ray candidate:  <A_0,X> + c^T x > 0
            ok_dunb = cand_dunb and hd <= par.infeastol * (-dobj)
            ok_dinf = cand_dinf and hp <= par.infeastol * pobj
            if (ok_dunb or ok_dinf) and (ok_dunb or not cand_dunb or certwait >= 5) \
                    and (ok_dinf or not cand_dinf or certwait >= 5):
                res.status = STATUS_PDINF if (ok_dunb and ok_dinf) else (STATUS_DUNB if ok_dunb else STATUS_DINF)
                break
            if ok_dunb or ok_dinf:
                certwait += 1
        if it == maxiter:
            break
        # stall: mu no longer decreases -> numerical limit reached
        if mu > 0.9 * lastmu and alpha_last < 1e-2:
            nstall += 1
            if nstall >= stall_lim:
                res.status = STATUS_NUMERIC
                break
        else:
            nstall = 0
        lastmu = mu
        # no progress: the worst scaled violation has not improved by 10 % for 6 iterations (accuracy limit of the problem)
        if not (tau < 1e-2 * min(1.0, kappa) or mu / (tau * tau) > 1e10):
            merit = max(pinf / par.feastol, dabs / par.feastol, gap / par.gaptol)
            if par.pabstol > 0.0:
                merit = max(merit, pabs / par.pabstol)
            if merit < 0.9 * bestmerit:
                bestmerit = merit
                sincebest = 0
            else:
                sincebest += 1
                if sincebest >= nobest_lim:
                    res.status = STATUS_NUMERIC
                    break

        # ---- factorizations and Schur complement
        try:
            Lz = [np.linalg.cholesky(Zk) for Zk in Z]
            Lx = [np.linalg.cholesky(Xk) for Xk in X]
        except np.linalg.LinAlgError:
            res.status = STATUS_NUMERIC
            break
        Zinv = []
        for L in Lz:
            Li = sla.solve_triangular(L, np.eye(L.shape[0]), lower=True, check_finite=False)
            Zinv.append(Li.T @ Li)
        Mx = np.zeros((m + 1, m + 1))
        if getattr(par, 'schur', 'U') == 'W':
            for A, Lxk, Lzk in zip(prob.blocks, Lx, Lz):
                Mx += schur_block_w(A, Lxk, Lzk)
        else:
            for A, Xk, Zi in zip(prob.blocks, X, Zinv):
                Mx += schur_block(A, Xk, Zi)
        if q:
            Mx += Dext.T @ ((x / z)[:, None] * Dext)
        g = Mx[0, 1:].copy()
        M = Mx[1:, 1:]
        Lm = chol_psd(M) if m else np.zeros((0, 0))

        def msolve(r):
            if m == 0:
                return r
            # each triangular solve is corrected once with the factor itself, as the engine's block solves are (chol.hip, hs_trsv
            # mode bit 4): the residual of M dy = h is the primal infeasibility the step leaves behind, and on nodes whose
            # optimum is not attained (tau -> 0) it decides whether the iteration still converges
            w = sla.solve_triangular(Lm, r, lower=True, check_finite=False)
            w = w + sla.solve_triangular(Lm, r - Lm @ w, lower=True, check_finite=False)
            v = sla.solve_triangular(Lm.T, w, lower=False, check_finite=False)
            return v + sla.solve_triangular(Lm.T, w - Lm.T @ v, lower=False, check_finite=False)

        # Stable elimination of (dtau, dkappa).  With w = M^-1 g the direction (1, -w) is the near-null direction of the
        # extended Schur matrix; omega - g^T M^-1 g is evaluated in factored form (a sum of non-negative terms) instead of
        # by subtraction, which loses all digits once x/z and X Z^-1 have spread over 16 orders of magnitude.
        w = msolve(g)
        ub = msolve(b)
        u2 = ub - w
        wt = np.concatenate([[1.0], -w])
        Bk = [np.tensordot(wt, A, axes=(0, 0)) for A in prob.blocks]      # A_0 - sum_i w_i A_i
        beta = Dext @ wt                                                   # c - D w
        S0 = sum(np.sum(Bm * (Xk @ Bm @ Zi)) for Bm, Xk, Zi in zip(Bk, X, Zinv)) + np.sum((x / z) * beta * beta)
        den = S0 + kappa / tau + b @ ub

        def direction(sigma, eta, E, e_lp, e_tk):
            """one Newton direction; E/e_lp/e_tk are the second-order terms dXa dZa, dxa*dza, dtau_a*dkappa_a"""
            H = []
            for k in range(K):
                G = eta * (X[k] @ Rd[k])
                if E is not None:
                    G = G + E[k]
                H.append(sigma * mu * Zinv[k] - X[k] - sym(G @ Zinv[k]))
            hl = sigma * mu / z - x - (eta * x * rd + (e_lp if e_lp is not None else 0.0)) / z
            AH = sum(Af @ Hk.reshape(-1) for Af, Hk in zip(Aflat, H)) if K else np.zeros(m + 1)
            AH = AH + Dext.T @ hl
            h = AH[1:] - eta * rp
            u1 = msolve(h)
            BH = sum(np.sum(Bm * Hk) for Bm, Hk in zip(Bk, H)) + beta @ hl
            num = -eta * rg + (sigma * mu - tau * kappa - e_tk) / tau - BH - eta * (w @ rp) + b @ u1
            dtau = num / den
            dy = u1 - u2 * dtau
            dyt = np.concatenate([[-dtau], dy])
            dZ = [np.tensordot(dyt, A, axes=(0, 0)) + eta * R for A, R in zip(prob.blocks, Rd)]
            dz = Dext @ dyt + eta * rd
            dX = []
            for k in range(K):
                G = X[k] @ dZ[k]
                if E is not None:
                    G = G + E[k]
                dX.append(sigma * mu * Zinv[k] - X[k] - sym(G @ Zinv[k]))
            dx = sigma * mu / z - x - (x * dz + (e_lp if e_lp is not None else 0.0)) / z
            dkappa = (sigma * mu - tau * kappa - e_tk - kappa * dtau) / tau
            return dy, dtau, dkappa, dX, dZ, dx, dz

        def steplen(dtau, dkappa, dX, dZ, dx, dz):
            a = np.inf
            for k in range(K):
                a = min(a, max_step_psd(Lx[k], dX[k]), max_step_psd(Lz[k], dZ[k]))
            a = min(a, max_step_vec(x, dx), max_step_vec(z, dz))
            if dtau < 0:
                a = min(a, -tau / dtau)
            if dkappa < 0:
                a = min(a, -kappa / dkappa)
            return a

        if not (np.isfinite(den) and np.all(np.isfinite(u2))):
            res.status = STATUS_NUMERIC
            break
        # ---- predictor
        dya, dta, dka, dXa, dZa, dxa, dza = direction(0.0, 1.0, None, None, 0.0)
        if not (np.isfinite(dta) and np.all(np.isfinite(dya)) and all(np.all(np.isfinite(d)) for d in dXa + dZa)):
            res.status = STATUS_NUMERIC
            break
        aa = min(1.0, steplen(dta, dka, dXa, dZa, dxa, dza))
        sigma = min(1.0, max(sigma_floor, (1.0 - aa) ** 3))
        eta = 1.0 - sigma
        # ---- corrector
        E = [dXa[k] @ dZa[k] for k in range(K)]
        dy, dt, dk, dX, dZ, dx, dz = direction(sigma, eta, E, dxa * dza, dta * dka)
        if not (np.isfinite(dt) and np.all(np.isfinite(dy)) and all(np.all(np.isfinite(d)) for d in dX + dZ)):
            res.status = STATUS_NUMERIC
            break
        amax = steplen(dt, dk, dX, dZ, dx, dz)
        alpha = min(1.0, gamma_eff * amax)
        alpha_last = alpha
        if not np.isfinite(alpha) or not np.all(np.isfinite(dy)):
            res.status = STATUS_NUMERIC
            break

        y = y + alpha * dy
        tau = tau + alpha * dt
        kappa = kappa + alpha * dk
        X = [sym(Xk + alpha * d) for Xk, d in zip(X, dX)]
        Z = [sym(Zk + alpha * d) for Zk, d in zip(Z, dZ)]
        x = x + alpha * dx
        z = z + alpha * dz

    res.iterations = it
    res.tau, res.kappa = tau, kappa
    res.raw = (y, X, Z, x, z, tau, kappa)
    if res.status in (STATUS_OPTIMAL, STATUS_ITERLIM, STATUS_NUMERIC):
        s = 1.0 / tau
    else:
        # rays are normalised by the objective value they certify
        s = 1.0 / max(abs(b @ y), abs(AX[0]), 1e-300)
    res.y = y * s
    res.X = [Xk * s for Xk in X]
    res.Z = [Zk * s for Zk in Z]
    res.x = x * s
    res.z = z * s
    res.pobj = float(AX[0] * s)
    res.dobj = float(b @ y * s)
    res.pinf, res.dinf, res.dabs, res.gap, res.mu = pinf, dinf, dabs, gap, mu
    return res
