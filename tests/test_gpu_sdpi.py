"""GPU tests at the drop-in boundary: the reference's SDPI known-answer tests (unittests/src/checksdpi.c) driven through
SCIPsdpiSolverLoadAndSolve[WithPenalty] of libhipsdp.so with arguments prepared like sdpi.c does (tests/harness/sdpi_prepare.py),
asserting what solveTest() asserts (checksdpi.c:125-364) at the same EPS = 1e-6."""
import ctypes as C
import json
import os
import numpy as np
import pytest

import sdpi_prepare
import sdpi_call
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))
EPS = CASES["eps"]


def build(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    return sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp)


def new_solver(gpu):
    s = sdpi_call.SdpiSolver(gpu.lib())
    assert s.set_real(3, EPS) == sdpi_call.SCIP_OKAY      # SDPSOLVERFEASTOL, checksdpi.c:96
    assert s.set_real(1, EPS) == sdpi_call.SCIP_OKAY      # GAPTOL, checksdpi.c:97
    return s


@pytest.mark.parametrize("case", CASES["cases"], ids=[c["name"] for c in CASES["cases"]])
def test_checksdpi_through_the_solver_interface(gpu, case):
    lib = gpu.lib()
    lib.hipsdp_compat_mem_used.restype = C.c_longlong
    base = lib.hipsdp_compat_mem_used()
    exp = case["expect"]
    tol = exp.get("tol", EPS)
    P = sdpi_prepare.prepare(build(case))
    s = new_solver(gpu)
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY
    assert s.flag("WasSolved") and not s.flag("IsObjlimExc") and not s.flag("IsIterlimExc") and not s.flag("IsTimelimExc")
    assert s.flag("IsAcceptable") and s.flag("FeasibilityKnown")
    rc, primalfeasible, dualfeasible = s.sol_feasibility()
    assert rc == sdpi_call.SCIP_OKAY
    pe, de = exp["primal"], exp["dual"]
    if pe == "feas" and de == "feas":
        assert s.flag("IsOptimal") and s.flag("IsDualFeasible") and not s.flag("IsDualInfeasible") and not s.flag("IsDualUnbounded")
        assert s.flag("IsPrimalFeasible") and not s.flag("IsPrimalInfeasible") and not s.flag("IsPrimalUnbounded")
        assert s.internal_status() == 0 and s.settings_used() == 1
    if pe == "feas":
        assert primalfeasible and s.flag("IsPrimalFeasible") and not s.flag("IsPrimalInfeasible") and not s.flag("IsDualUnbounded")
    elif pe == "unbounded":
        assert not s.flag("IsPrimalInfeasible") and not s.flag("IsDualFeasible") and s.flag("IsDualInfeasible")
    elif pe == "infeas":
        assert not primalfeasible and not s.flag("IsPrimalFeasible") and not s.flag("IsPrimalUnbounded")
    if de == "feas":
        assert dualfeasible and s.flag("IsDualFeasible") and not s.flag("IsDualInfeasible") and not s.flag("IsPrimalUnbounded")
    elif de == "unbounded":
        assert not s.flag("IsPrimalFeasible") and s.flag("IsPrimalInfeasible")
    elif de == "infeas":
        assert not dualfeasible and not s.flag("IsDualFeasible") and not s.flag("IsDualUnbounded")
    if "dualsol" in exp:
        rc, objval, y = s.dual_sol()
        assert rc == sdpi_call.SCIP_OKAY
        assert np.allclose(y, exp["dualsol"], atol=tol)
        assert abs(objval - exp["objval"]) <= 10 * tol
    rc, lbv, ubv = s.bound_vars()
    assert rc == sdpi_call.SCIP_OKAY
    rc, lhs, rhs = s.lp_sides()
    assert rc == sdpi_call.SCIP_OKAY
    lhsm, rhsm = sdpi_prepare.map_lp_sides(P, lhs, rhs, lbv, ubv)          # sdpi.c:4473-4605
    if "lbvals" in exp:
        assert np.allclose(lbv, exp["lbvals"], atol=tol)
    if "lhsvals" in exp:
        assert np.allclose(lhsm, exp["lhsvals"], atol=tol)
    if "rhsvals" in exp:
        assert np.allclose(rhsm, exp["rhsvals"], atol=tol)
    if "X" in exp:
        rc, mats = s.primal_solution_matrix()
        assert rc == sdpi_call.SCIP_OKAY
        assert np.allclose(mats[0], exp["X"], atol=10 * tol)
        rc, sparse = s.primal_matrix_sparse()
        assert rc == sdpi_call.SCIP_OKAY
        dense = np.zeros_like(mats[0])
        r, c, v = sparse[0]
        dense[r, c] = v
        dense[c, r] = v
        assert np.allclose(dense, mats[0], atol=2e-9)        # entries below epsilon = 1e-9 are dropped
        assert np.all(r >= c)
    assert s.iterations() > 0 and s.sdpcalls() >= 1 and s.opttime() > 0
    s.free()
    assert lib.hipsdp_compat_mem_used() == base


def test_fixed_variables_and_removed_indices(gpu):
    """a node after branching: one variable fixed, one block index emptied, one block removed completely; X must come
    back in ORIGINAL indices with zeros at the removed ones (sdpisolver_dsdp.c:2467-2545)"""
    blocks = [dict(n=3, vars={0: [(0, 0, 1.0)], 1: [(1, 1, 1.0)], 2: [(2, 2, 1.0)]}, const=[(0, 0, 1.0), (1, 1, 2.0)]),
              dict(n=2, vars={2: [(0, 0, 1.0), (1, 1, 1.0)]}, const=[])]
    prob = sdpi_prepare.SdpiProblem([1, 1, 1], [-1e20, -1e20, 0.0], [1e20, 1e20, 0.0], blocks, [])   # y3 fixed to 0
    P = sdpi_prepare.prepare(prob)
    assert list(P.indchanges[0]) == [0, 0, -1] and P.blockindchanges == [0, -1]
    s = new_solver(gpu)
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    rc, obj, y = s.dual_sol()
    assert np.allclose(y, [1.0, 2.0, 0.0], atol=1e-5) and abs(obj - 3.0) <= 1e-5
    rc, mats = s.primal_solution_matrix()
    assert np.allclose(mats[0], np.diag([1.0, 1.0, 0.0]), atol=1e-5)
    assert np.all(mats[1] == 0.0)
    s.free()


def test_penalty_formulation(gpu):
    """sdpi.c:3452 / :3521 call patterns: (Gamma = 1, withobj = F, rbound = F) feasibility problem and (Gamma, T, T)"""
    case = [c for c in CASES["cases"] if c["name"] == "test10"][0]
    P = sdpi_prepare.prepare(build(case))
    s = new_solver(gpu)
    rc, feasorig, pbound = s.solve(P, penaltyparam=1.0, withobj=False, rbound=False)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsAcceptable")
    rc, obj, y = s.dual_sol()
    assert obj <= 1e-5            # a feasible problem has a non-positive penalty optimum (sdpisolver.h:246-248)
    rc, feasorig, pbound = s.solve(P, penaltyparam=1e5, withobj=True, rbound=True)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    assert feasorig and s.settings_used() == 0
    rc, obj, y = s.dual_sol()
    assert np.allclose(y, [1, 1], atol=1e-4) and abs(obj + 2.0) <= 1e-4
    # infeasible problem (test9): r stays positive, the feasibility problem has a positive optimum
    case9 = [c for c in CASES["cases"] if c["name"] == "test9"][0]
    P9 = sdpi_prepare.prepare(build(case9))
    rc, feasorig, pbound = s.solve(P9, penaltyparam=1.0, withobj=False, rbound=False)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal") and not feasorig
    rc, obj, y = s.dual_sol()
    assert obj > 1e-3
    s.free()


def _random_node(rng, nvars, sizes, nlp, nfixed):
    """a random branch-and-bound-node-like problem: bounded variables, some fixed, sparse blocks, ranged LP rows"""
    lb = -rng.uniform(0.5, 2.0, nvars)
    ub = rng.uniform(0.5, 2.0, nvars)
    for v in rng.choice(nvars, nfixed, replace=False):
        val = float(rng.integers(-1, 2))
        lb[v] = ub[v] = val
    blocks = []
    for n in sizes:
        vars_ = {}
        for v in range(nvars):
            if rng.random() < 0.7:
                ents = []
                for _ in range(rng.integers(1, 4)):
                    r = int(rng.integers(0, n)); c = int(rng.integers(0, r + 1))
                    ents.append((r, c, float(rng.standard_normal())))
                ents = list({(r, c): (r, c, x) for (r, c, x) in ents}.values())
                vars_[v] = ents
        const = [(i, i, -float(rng.uniform(2.0, 4.0))) for i in range(n)]        # A_0 = -dI: y = 0 is strictly feasible
        blocks.append(dict(n=n, vars=vars_, const=const))
    lp = []
    for _ in range(nlp):
        k = int(rng.integers(1, min(4, nvars) + 1))
        idx = rng.choice(nvars, k, replace=False)
        row = {int(v): float(rng.standard_normal()) for v in idx}
        lhs = -float(rng.uniform(1.0, 3.0)) if rng.random() < 0.7 else -1e20
        rhs = float(rng.uniform(1.0, 3.0)) if (rng.random() < 0.7 or lhs == -1e20) else 1e20
        lp.append((lhs, rhs, row))
    return sdpi_prepare.SdpiProblem(rng.standard_normal(nvars), lb, ub, blocks, lp)


@pytest.mark.parametrize("seed", range(6))
def test_random_mid_size_nodes_backend_vs_oracle(gpu, seed):
    """the same end-to-end comparison with blocks of 66-150 rows and 70-200 variables: fixings, removed rows / columns and LP rows
    through the general kernels (packed passes, blocked factorizations, MFMA tile products, split-K slabs shared by two blocks)"""
    import ipm_ref
    import checker
    rng = np.random.default_rng(7000 + seed)
    prob = _random_node(rng, nvars=int(rng.integers(70, 200)), sizes=[int(rng.integers(66, 150)), int(rng.integers(20, 110))],
                        nlp=int(rng.integers(0, 40)), nfixed=int(rng.integers(0, 12)))
    P = sdpi_prepare.prepare(prob)
    if P.status != 'ok':
        pytest.skip("presolve decided the node")
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    core = ipm_ref.CoreProblem(b, blk, D, c)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    s = new_solver(gpu)
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY
    if ref.status == ipm_ref.STATUS_DINF:               # the random LP rows can contradict each other: same verdict expected
        assert s.flag("IsDualInfeasible") and not s.flag("IsOptimal")
        s.free()
        return
    assert ref.status == ipm_ref.STATUS_OPTIMAL and s.flag("IsOptimal")
    rc, objval, y = s.dual_sol()
    fixedcontr = sum(prob.obj[v] * P.lb[v] for v in range(prob.nvars) if v not in maps["active"])
    assert abs(objval - (ref.dobj + fixedcontr)) <= 1e-5 * (1 + abs(objval))
    yact = np.array([y[v] for v in maps["active"]])
    assert checker.check_dual(core, yact, 1e-5)["feasible"]
    s.free()


@pytest.mark.parametrize("seed", range(8))
def test_random_nodes_backend_vs_oracle(gpu, seed):
    """end to end at the boundary: sdpi-style preparation -> SCIPsdpiSolverLoadAndSolve (HIP) against the independently
    marshalled problem (sdpi_prepare.to_core) solved by the oracle; objective to 1e-5, y feasible by the LAPACK checker"""
    import ipm_ref
    import checker
    rng = np.random.default_rng(1000 + seed)
    prob = _random_node(rng, nvars=int(rng.integers(4, 12)), sizes=[int(rng.integers(2, 7)), int(rng.integers(2, 9))],
                        nlp=int(rng.integers(0, 6)), nfixed=int(rng.integers(0, 3)))
    P = sdpi_prepare.prepare(prob)
    if P.status != 'ok':
        pytest.skip("presolve decided the node")
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    core = ipm_ref.CoreProblem(b, blk, D, c)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    s = new_solver(gpu)
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY
    assert ref.status == ipm_ref.STATUS_OPTIMAL and s.flag("IsOptimal")
    rc, objval, y = s.dual_sol()
    fixedcontr = sum(prob.obj[v] * P.lb[v] for v in range(prob.nvars) if v not in maps["active"])
    assert abs(objval - (ref.dobj + fixedcontr)) <= 1e-5 * (1 + abs(objval))
    yact = np.array([y[v] for v in maps["active"]])
    assert checker.check_dual(core, yact, 1e-5)["feasible"]
    for v in range(prob.nvars):
        if v not in maps["active"]:
            assert y[v] == P.lb[v]
    s.free()


def test_termination_bound_against_the_independent_check_of_y(gpu, monkeypatch):
    """when the engine's own dual residual bound settles the caller's tolerance (dabs <= 0.999 feastol) the backend
    skips the eigenvalue check of Z(y).  HIPSDP_VERIFY_SHORTCUT=1 evaluates hipsdp_check_y_tol all the same and fails the solve when
    lambda_min(Z(y)) < -dabs or an LP row is violated by more than dabs: the golden cases and forty random nodes (one-launch kernel and
    general path) pass through it, and the shortcut did fire."""
    lib = gpu.lib()
    lib.hipsdp_compat_shortcut_checks.restype = C.c_longlong
    before = lib.hipsdp_compat_shortcut_checks()
    monkeypatch.setenv("HIPSDP_VERIFY_SHORTCUT", "1")
    for case in CASES["cases"]:
        P = sdpi_prepare.prepare(build(case))
        if P.status != 'ok':
            continue
        s = new_solver(gpu)
        rc, _, _ = s.solve(P)
        assert rc == sdpi_call.SCIP_OKAY, case["name"]
        s.free()
    done = 0
    for path in ("1", "0"):
        monkeypatch.setenv("HIPSDP_SOLVE1", path)
        rng = np.random.default_rng(31337)
        for t in range(20):
            big = (t % 4 == 3)
            prob = _random_node(rng, nvars=int(rng.integers(30, 80) if big else rng.integers(4, 12)),
                                sizes=[int(rng.integers(66, 100))] if big else [int(rng.integers(2, 7)), int(rng.integers(2, 9))],
                                nlp=int(rng.integers(0, 6)), nfixed=int(rng.integers(0, 3)))
            P = sdpi_prepare.prepare(prob)
            if P.status != 'ok':
                continue
            s = new_solver(gpu)
            rc, _, _ = s.solve(P)
            assert rc == sdpi_call.SCIP_OKAY
            done += 1
            s.free()
    monkeypatch.delenv("HIPSDP_SOLVE1", raising=False)
    assert done >= 20
    assert lib.hipsdp_compat_shortcut_checks() >= before + 20


def test_time_limit_and_objective_limit(gpu):
    case = [c for c in CASES["cases"] if c["name"] == "test11"][0]
    P = sdpi_prepare.prepare(build(case))
    s = new_solver(gpu)
    assert s.set_real(4, 2.0) == sdpi_call.SCIP_OKAY             # OBJLIMIT below the optimum 5: the lower bound passes it
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("WasSolved")
    assert s.flag("IsObjlimExc") and s.flag("IsAcceptable") and not s.flag("IsOptimal") and s.internal_status() == 3
    assert s.set_real(4, 1e20) == sdpi_call.SCIP_OKAY
    rc, _, _ = s.solve(P)
    assert s.flag("IsOptimal") and not s.flag("IsObjlimExc")
    s.free()


def test_master_copy_is_reused_and_invalidated(gpu):
    """the device-resident master copy of the A_v: node 2 (other fixings, same arrays) re-uses it,
    a problem with one changed coefficient must not"""
    import time
    import ipm_ref
    rng = np.random.default_rng(77)
    nvars, n = 30, 40
    vars_ = {}
    for v in range(nvars):
        G = rng.standard_normal((n, n))
        ents = [(r, c, float(G[r, c] + G[c, r])) for r in range(n) for c in range(r + 1)]
        vars_[v] = ents
    const = [(i, i, -5.0) for i in range(n)]

    def make(lb, ub, blocks):
        return sdpi_prepare.SdpiProblem(np.linspace(-1, 1, nvars), lb, ub, blocks, [])

    blocks = make(-np.ones(nvars), np.ones(nvars), [dict(n=n, vars=vars_, const=const)]).blocks     # normalised once, shared
    lb = -np.ones(nvars); ub = np.ones(nvars)
    s = new_solver(gpu)

    def run(prob):
        P = sdpi_prepare.prepare(prob)
        t = time.perf_counter()
        rc, _, _ = s.solve(P)
        dt = time.perf_counter() - t
        assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
        rc, obj, y = s.dual_sol()
        b, blk, D, c, maps = sdpi_prepare.to_core(P)
        ref = ipm_ref.hsd_solve(ipm_ref.CoreProblem(b, blk, D, c), ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
        fixedcontr = sum(prob.obj[v] * P.lb[v] for v in range(prob.nvars) if v not in maps["active"])
        assert abs(obj - (ref.dobj + fixedcontr)) <= 1e-5 * (1 + abs(obj))
        return dt, obj

    t1, o1 = run(make(lb, ub, blocks))
    lb2, ub2 = lb.copy(), ub.copy()
    lb2[3] = ub2[3] = 0.0
    lb2[7] = ub2[7] = 0.0
    lb2[11] = ub2[11] = 0.1
    t2, o2 = run(make(lb2, ub2, blocks))                   # same arrays, other fixings: master copy re-used
    vars3 = dict(vars_)
    vars3[5] = [(r, c, x * (2.0 if (r, c) == (0, 0) else 1.0)) for (r, c, x) in vars_[5]]
    t3, o3 = run(make(lb, ub, [dict(n=n, vars=vars3, const=const)]))   # one changed value: fingerprint must differ
    print("master copy: first call %.4f s (upload), second call %.4f s (re-used), changed data %.4f s" % (t1, t2, t3))
    s.free()


def test_warm_start_through_the_solver_interface(gpu):
    """starty / startZ / startX of SCIPsdpiSolverLoadAndSolve (sdpisolver.h:160-173): an interior point near the optimum of
    checksdpi test11 (min x, x I - [1 2; 2 4] psd: y = 5, X = [0.2 0.4; 0.4 0.8]) saves iterations; entries of the diagonal LP
    block with the 2 * nlpcons + 2 * var (+1) convention are mapped to the bound rows (test10-like problem with bounds)."""
    case = [c for c in CASES["cases"] if c["name"] == "test11"][0]
    P = sdpi_prepare.prepare(build(case))
    s = new_solver(gpu)
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    cold_it = s.iterations()
    y0 = 5.05
    Z0 = np.array([[y0 - 1.0, -2.0], [-2.0, y0 - 4.0]])
    X0 = np.array([[0.2, 0.4], [0.4, 0.8]]) + 0.01 * np.eye(2)
    assert np.linalg.eigvalsh(Z0)[0] > 0
    low = [(0, 0), (1, 0), (1, 1)]
    start = dict(y=[y0],
                 Z=[([r for r, c in low], [c for r, c in low], [Z0[r, c] for r, c in low]), ([], [], [])],
                 X=[([r for r, c in low], [c for r, c in low], [X0[r, c] for r, c in low]), ([], [], [])])
    rc, _, _ = s.solve(P, start=start)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    rc, obj, y = s.dual_sol()
    assert abs(y[0] - 5.0) <= 1e-5
    assert s.iterations() < cold_it
    # a start that is not interior (singular X) is ignored: same iteration count as the cold start
    startbad = dict(start)
    Xs = np.array([[0.2, 0.4], [0.4, 0.8]])
    startbad["X"] = [([r for r, c in low], [c for r, c in low], [Xs[r, c] for r, c in low]), ([], [], [])]
    rc, _, _ = s.solve(P, start=startbad)
    assert s.flag("IsOptimal") and s.iterations() == cold_it
    s.free()


def test_preoptimal_solution_through_the_solver_interface(gpu):
    """SCIPsdpiSolverGetPreoptimalSol / GetPreoptimalPrimalNonzeros (sdpisolver.h; SDPA semantics sdpisolver_sdpa.cpp:2426-2670):
    with SCIP_SDPPAR_WARMSTARTPOGAP set the backend returns the preoptimal y in original indices and X in the sparse
    original-index format with the LP block last; without it success = FALSE and the first count is -1"""
    case = [c for c in CASES["cases"] if c["name"] == "test11"][0]          # min x, x I - [1 2; 2 4] psd: y = 5, X = [.2 .4; .4 .8]
    P = sdpi_prepare.prepare(build(case))
    s = new_solver(gpu)
    PI, PD = C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib = gpu.lib()

    def ask():
        nb = len(P.prob.blocks) + 1
        cnt = np.zeros(nb, dtype=np.int32)
        assert lib.SCIPsdpiSolverGetPreoptimalPrimalNonzeros(s.h, nb, cnt.ctypes.data_as(PI)) == sdpi_call.SCIP_OKAY
        if cnt[0] < 0:
            return None
        rows = [np.zeros(max(int(k), 1), dtype=np.int32) for k in cnt]
        cols = [np.zeros(max(int(k), 1), dtype=np.int32) for k in cnt]
        vals = [np.zeros(max(int(k), 1)) for k in cnt]
        pr, pc, pv = (PI * nb)(), (PI * nb)(), (PD * nb)()
        for b in range(nb):
            pr[b], pc[b], pv[b] = rows[b].ctypes.data_as(PI), cols[b].ctypes.data_as(PI), vals[b].ctypes.data_as(PD)
        ok = C.c_uint(0)
        y = np.zeros(P.prob.nvars)
        cnt2 = cnt.copy()
        assert lib.SCIPsdpiSolverGetPreoptimalSol(s.h, C.byref(ok), y.ctypes.data_as(PD), nb, cnt2.ctypes.data_as(PI), pr, pc, pv) \
            == sdpi_call.SCIP_OKAY
        assert ok.value == 1 and list(cnt2) == list(cnt)
        return y, [(rows[b][:cnt[b]], cols[b][:cnt[b]], vals[b][:cnt[b]]) for b in range(nb)]

    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    assert ask() is None                                                       # parameter not set (default -1)
    assert s.set_real(12, 1e-2) == sdpi_call.SCIP_OKAY                         # SCIP_SDPPAR_WARMSTARTPOGAP
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    got = ask()
    assert got is not None
    y, X = got
    rc, obj, yfin = s.dual_sol()
    assert abs(y[0] - 5.0) <= 0.2      # near the optimum (on this tiny problem the first feasible iterate may be the last one)
    r, c, v = X[0]
    M = np.zeros((2, 2))
    for rr, cc, vv in zip(r, c, v):
        M[rr, cc] = vv
        M[cc, rr] = vv
    assert np.linalg.eigvalsh(M)[0] > 0.0 and np.max(np.abs(M - np.array([[0.2, 0.4], [0.4, 0.8]]))) <= 0.1
    s.free()


def test_deferred_setters_give_the_results_of_the_direct_ones(gpu, monkeypatch):
    """A sequence of different random nodes through ONE backend (shapes change from node to node: buffers allocated with room are
    re-used, the setters of a node wait as commands in the pinned arena for the launch of the solve) against a second backend with
    HIPSDP_NO_STAGING=1 and HIPSDP_NO_SHAPE_REUSE=1 (every setter copies and waits, every shape allocates anew): same status, same
    iteration count, objective and y to the last bit - the arithmetic of the solve does not depend on how its data arrived."""
    sa, sb = new_solver(gpu), new_solver(gpu)
    rng = np.random.default_rng(77)
    done = 0
    for t in range(30):
        prob = _random_node(rng, nvars=int(rng.integers(4, 12)), sizes=[int(rng.integers(2, 7)), int(rng.integers(2, 9))],
                            nlp=int(rng.integers(0, 6)), nfixed=int(rng.integers(0, 3)))
        P = sdpi_prepare.prepare(prob)
        if P.status != 'ok':
            continue
        monkeypatch.delenv("HIPSDP_NO_STAGING", raising=False)
        monkeypatch.delenv("HIPSDP_NO_SHAPE_REUSE", raising=False)
        rca, _, _ = sa.solve(P)
        monkeypatch.setenv("HIPSDP_NO_STAGING", "1")
        monkeypatch.setenv("HIPSDP_NO_SHAPE_REUSE", "1")
        rcb, _, _ = sb.solve(P)
        assert rca == rcb == sdpi_call.SCIP_OKAY
        assert sa.flag("IsOptimal") == sb.flag("IsOptimal") and sa.flag("IsDualInfeasible") == sb.flag("IsDualInfeasible")
        assert sa.iterations() == sb.iterations()
        if sa.flag("IsOptimal"):
            _, oa, ya = sa.dual_sol()
            _, ob, yb = sb.dual_sol()
            assert oa == ob and np.array_equal(ya, yb)
        done += 1
    monkeypatch.delenv("HIPSDP_NO_STAGING", raising=False)
    monkeypatch.delenv("HIPSDP_NO_SHAPE_REUSE", raising=False)
    sa.free(); sb.free()
    assert done >= 15


def test_deferred_setters_with_operands_above_the_command_limit(gpu, monkeypatch):
    """a gather of nactive * nkept^2 > NC_LIMIT (65536) entries and the clearing of more than NC_LIMIT doubles flush the
    waiting command list in the MIDDLE of a node's setters (objective copy pending -> flush -> further commands into slot 0).  The
    launch reads the list when it runs: the next command must not be written before it has.  Blocks of 44-60 rows with 34-48
    variables (one-launch path declines nothing here that matters: the data path is the same), a second small block so that the
    shape re-use path clears mixed sizes; against a backend with every setter blocking: same bits."""
    sa, sb = new_solver(gpu), new_solver(gpu)
    rng = np.random.default_rng(4711)
    done = 0
    for t in range(12):
        prob = _random_node(rng, nvars=int(rng.integers(34, 48)), sizes=[int(rng.integers(44, 60)), int(rng.integers(2, 9))],
                            nlp=int(rng.integers(0, 6)), nfixed=int(rng.integers(0, 3)))
        P = sdpi_prepare.prepare(prob)
        if P.status != 'ok':
            continue
        monkeypatch.delenv("HIPSDP_NO_STAGING", raising=False)
        monkeypatch.delenv("HIPSDP_NO_SHAPE_REUSE", raising=False)
        rca, _, _ = sa.solve(P)
        monkeypatch.setenv("HIPSDP_NO_STAGING", "1")
        monkeypatch.setenv("HIPSDP_NO_SHAPE_REUSE", "1")
        rcb, _, _ = sb.solve(P)
        assert rca == rcb == sdpi_call.SCIP_OKAY
        assert sa.flag("IsOptimal") == sb.flag("IsOptimal") and sa.flag("IsDualInfeasible") == sb.flag("IsDualInfeasible")
        assert sa.iterations() == sb.iterations()
        if sa.flag("IsOptimal"):
            _, oa, ya = sa.dual_sol()
            _, ob, yb = sb.dual_sol()
            assert oa == ob and np.array_equal(ya, yb)
        done += 1
    monkeypatch.delenv("HIPSDP_NO_STAGING", raising=False)
    monkeypatch.delenv("HIPSDP_NO_SHAPE_REUSE", raising=False)
    sa.free(); sb.free()
    assert done >= 6
