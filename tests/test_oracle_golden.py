"""CPU tests: the oracle (oracle/ipm_ref.py + tests/harness/sdpi_prepare.py) against the reference's known answers and against
algorithm-independent certificates.  This is what pins the oracle (prompt section 3): if these fail, no GPU parity claim
means anything."""
import json
import os
import numpy as np
import pytest

import ipm_ref
import sdpi_prepare
import sdpa_io
import checker
import instances
from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))
EPS = CASES["eps"]


def build(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    return sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp)


def solve_case(case, tol=1e-7):
    P = sdpi_prepare.prepare(build(case))
    b, blocks, D, c, maps = sdpi_prepare.to_core(P)
    core = ipm_ref.CoreProblem(b, blocks, D, c)
    res = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=tol, feastol=tol))
    return P, core, maps, res


EXPECTED_STATUS = {("feas", "feas"): (ipm_ref.STATUS_OPTIMAL,),
                   ("infeas", "unbounded"): (ipm_ref.STATUS_DUNB,),
                   ("unbounded", "infeas"): (ipm_ref.STATUS_DINF,),
                   ("infeas", "infeas"): (ipm_ref.STATUS_PDINF,)}


@pytest.mark.parametrize("case", CASES["cases"], ids=[c["name"] for c in CASES["cases"]])
def test_checksdpi_known_answers(case):
    P, core, maps, res = solve_case(case)
    exp = case["expect"]
    assert res.status in EXPECTED_STATUS[(exp["primal"], exp["dual"])], (res.status, exp)
    tol = exp.get("tol", EPS)
    if "dualsol" in exp:
        y = np.array(P.lb, dtype=float)
        for k, v in enumerate(maps["active"]):
            y[v] = res.y[k]
        assert np.allclose(y, exp["dualsol"], atol=tol)
        assert abs(float(np.dot(case["obj"], y)) - exp["objval"]) <= 10 * tol
    # multipliers: engine rows are [LP sides..., bounds...] in the order of to_core
    nsides = sum((1 if P.lpindchanges[r] >= 0 and P.lplhs[r] > -1e20 else 0) + (1 if P.lpindchanges[r] >= 0 and P.lprhs[r] < 1e20 else 0)
                 for r in range(P.nlpcons))
    lbvals = np.zeros(P.prob.nvars)
    ubvals = np.zeros(P.prob.nvars)
    pos = nsides
    for v in maps["active"]:
        if P.lb[v] > -1e20:
            lbvals[v] = res.x[pos]
            pos += 1
        if P.ub[v] < 1e20:
            ubvals[v] = res.x[pos]
            pos += 1
    lhsv = np.zeros(P.nlpcons)
    rhsv = np.zeros(P.nlpcons)
    pos = 0
    for r in range(P.nlpcons):
        if P.lpindchanges[r] < 0:
            continue
        if P.lplhs[r] > -1e20:
            lhsv[r] = res.x[pos]
            pos += 1
        if P.lprhs[r] < 1e20:
            rhsv[r] = res.x[pos]
            pos += 1
    lhsm, rhsm = sdpi_prepare.map_lp_sides(P, lhsv, rhsv, lbvals, ubvals)
    if "lbvals" in exp:
        assert np.allclose(lbvals, exp["lbvals"], atol=tol)
    if "lhsvals" in exp:
        assert np.allclose(lhsm, exp["lhsvals"], atol=tol)
    if "rhsvals" in exp:
        assert np.allclose(rhsm, exp["rhsvals"], atol=tol)
    if "X" in exp:
        assert np.allclose(res.X[0], exp["X"], atol=10 * tol)
    # algorithm-independent certificates
    if res.status == ipm_ref.STATUS_OPTIMAL:
        ok, det = checker.certificate(core, res.y, res.X, res.x, 1e-5, 1e-5)
        assert ok, det
    if res.status in (ipm_ref.STATUS_DINF, ipm_ref.STATUS_PDINF):
        ok, det = checker.farkas_dual_infeasible(core, res.X, res.x, 1e-6)
        assert ok, det
    if res.status in (ipm_ref.STATUS_DUNB, ipm_ref.STATUS_PDINF):
        ok, det = checker.farkas_dual_unbounded(core, res.y, 1e-6)
        assert ok, det


# MISDP optima of check/testset/short.solu: the root relaxation must be a lower bound (all instances are minimisations)
SOLU = {"example_small.dat-s": -8.0, "example_TT.dat-s.gz": 2.11803, "example_CLS.dat-s.gz": 7.1485,
        "example_MkP.dat-s.gz": -95.0, "example_tightenmatrices.dat-s": -9.0}


@pytest.mark.parametrize("name", sorted(SOLU))
def test_root_relaxation_bounds_misdp_optimum(name):
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", name))
    D, c = sdpa_io.lp_dense(inst)
    core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
    res = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    assert res.status == ipm_ref.STATUS_OPTIMAL
    ok, det = checker.certificate(core, res.y, res.X, res.x, 1e-5, 1e-5)
    assert ok, det
    assert res.dobj <= SOLU[name] + 1e-4


def test_example_small_integer_point_is_feasible_with_value_minus8():
    """example_small has the MISDP optimum -8 (short.solu:1): find the best integer point by enumeration over the box the
    LP block defines and confirm value and feasibility with the checker - pins reader + checker on a known optimum."""
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", "example_small.dat-s"))
    D, c = sdpa_io.lp_dense(inst)
    core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
    best = None
    rng = range(-10, 11)
    for y1 in rng:
        for y2 in rng:
            for y3 in rng:
                y = np.array([y1, y2, y3], dtype=float)
                if checker.check_dual(core, y, 1e-9)["feasible"]:
                    v = float(inst.obj @ y)
                    if best is None or v < best:
                        best = v
    assert best == pytest.approx(-8.0)


@pytest.mark.parametrize("n,m", [(5, 8), (20, 40), (40, 60)])
def test_planted_optimum(n, m):
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    assert abs(b @ ys - np.sum(A[0] * Xs)) < 1e-9 * max(1.0, abs(b @ ys))     # planted pair has zero gap
    core = ipm_ref.CoreProblem(b, [A])
    res = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    assert res.status == ipm_ref.STATUS_OPTIMAL
    assert abs(res.dobj - b @ ys) <= 1e-5 * (1 + abs(b @ ys))
    ok, det = checker.certificate(core, res.y, res.X, res.x, 1e-5, 1e-5)
    assert ok, det


def test_lapack_wrapper_golden_gemm():
    """unittests/src/checklapack.c:80-119: [1 3; 2 4] * [5 7; 6 8]^T = [26 30; 38 44] in column-major storage"""
    A = np.array([1, 2, 3, 4], dtype=float).reshape(2, 2).T      # column-major {1,2,3,4}
    B = np.array([5, 6, 7, 8], dtype=float).reshape(2, 2).T
    Cm = A @ B.T
    assert np.allclose(Cm.T.reshape(-1), [26, 38, 30, 44])


def test_schur_block_matches_einsum():
    rng = np.random.default_rng(3)
    m1, n = 7, 6
    A = rng.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    G = rng.standard_normal((n, n)); X = G @ G.T + np.eye(n)
    G = rng.standard_normal((n, n)); Zi = np.linalg.inv(G @ G.T + np.eye(n))
    ref = np.einsum('iab,bc,jcd,da->ij', A, X, A, Zi)
    assert np.allclose(ipm_ref.schur_block(A, X, Zi), 0.5 * (ref + ref.T), rtol=1e-12, atol=1e-12)


def test_oracle_warm_start_rule():
    """warm_start_point accepts strictly interior points only; a warm start near the optimum needs fewer iterations"""
    import instances
    b, A, ys, Xs, Zs = instances.planted_dense(12, 18)
    core = ipm_ref.CoreProblem(b, [A])
    cold = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    assert ipm_ref.warm_start_point(core, ys, [Xs], [Zs], np.zeros(0), np.zeros(0)) is None      # singular pair
    d = 1e-2
    st = ipm_ref.warm_start_point(core, cold.y, [(1 - d) * cold.X[0] + d * np.eye(12)], [(1 - d) * cold.Z[0] + d * np.eye(12)],
                                  np.zeros(0), np.zeros(0))
    assert st is not None and st[5] == 1.0 and st[6] > 0
    warm = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6), start=st)
    assert warm.status == cold.status == ipm_ref.STATUS_OPTIMAL
    assert warm.iterations < cold.iterations
    assert abs(warm.dobj - cold.dobj) <= 1e-6 * (1 + abs(cold.dobj))


def test_eigenvector_cut_restatement_matches_the_sparse_formula():
    """oracle/eigcuts_ref.py: the dense coefficients v^T A_j v equal the reference's sparse accumulation
    (cons_sdp.c:826-865: off-diagonal entries of the lower triangle counted twice)"""
    import eigcuts_ref
    rng = np.random.default_rng(4)
    n, m = 6, 4
    A = rng.standard_normal((m + 1, n, n))
    A = A + A.transpose(0, 2, 1)
    y = rng.standard_normal(m)
    ev, co, lh, ve = eigcuts_ref.cuts_dense(A, y, 1e-9, 3)
    assert len(ev) >= 1
    for c, v in enumerate(ve):
        for i in range(m):
            ents = [(r, cc, A[1 + i, r, cc]) for r in range(n) for cc in range(r + 1)]
            assert abs(eigcuts_ref.vAv_sparse(ents, v) - co[c, i]) <= 1e-12 * max(1.0, abs(co[c, i]))
        assert abs((co[c] @ y - lh[c]) - ev[c]) <= 1e-10


def test_pair_formula_over_nonzeros_equals_the_dense_schur_formula():
    """oracle/ipm_ref.schur_pairs_sparse (SDPA's F3 case) and schur_rows_sparse (the two-stage sum of csrc/sparse.hip) against the
    three-product formula on the dense expansion of the same matrices"""
    import instances
    rng = np.random.default_rng(4)
    for n, m, k in ((12, 9, 1), (30, 25, 3), (41, 17, 20)):
        b, coo, A0, ys, Xs, Zs = instances.planted_sparse(n, m, k, seed=n + m)
        X = rng.standard_normal((n, n)); X = X @ X.T + n * np.eye(n)
        Z = rng.standard_normal((n, n)); Z = Z @ Z.T + n * np.eye(n)
        Zi = np.linalg.inv(Z)
        M1 = ipm_ref.schur_block(instances.coo_to_dense(n, m, coo, A0), X, Zi)[1:, 1:]
        M2 = ipm_ref.schur_pairs_sparse(m, coo, X, Zi)
        assert np.max(np.abs(M1 - M2)) <= 1e-12 * np.max(np.abs(M1))
        M3 = ipm_ref.schur_rows_sparse(m, n, coo, X, Zi)             # the association csrc/sparse.hip uses since round 4
        assert np.max(np.abs(M1 - M3)) <= 1e-12 * np.max(np.abs(M1))
        # and the planted optimum is what the oracle finds on the expansion
        core = ipm_ref.CoreProblem(b, [instances.coo_to_dense(n, m, coo, A0)])
        r = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-7, feastol=1e-7))
        if k > 1:
            assert r.status == 0 and abs(r.dobj - b @ ys) <= 1e-6 * (1 + abs(b @ ys))


def test_w_formulation_of_the_cpu_baseline_equals_the_default_schur_formula():
    rng = np.random.default_rng(2)
    n, m1 = 37, 23
    A = rng.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    X = rng.standard_normal((n, n)); X = X @ X.T + n * np.eye(n)
    Z = rng.standard_normal((n, n)); Z = Z @ Z.T + n * np.eye(n)
    M1 = ipm_ref.schur_block(A, X, np.linalg.inv(Z))
    M2 = ipm_ref.schur_block_w(A, np.linalg.cholesky(X), np.linalg.cholesky(Z))
    assert np.max(np.abs(M1 - M2)) <= 1e-12 * np.max(np.abs(M1))
