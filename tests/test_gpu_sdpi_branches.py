"""Branches of the solver interface that the known-answer tests do not reach (the settings
ladder of the backend, re-entrancy): each one is forced through the public SCIPsdpiSolver* surface of libhipsdp.so."""
import threading
import json
import os
import numpy as np
import pytest

import sdpi_prepare
import sdpi_call
import sdpi_driver as drv
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))


def build(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    return sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp)


def case(name):
    return [c for c in CASES["cases"] if c["name"] == name][0]


def test_tolerance_loop_takes_a_second_engine_solve(gpu):
    """row a5 (sdpisolver_dsdp.c:1527-1606, checkFeastolAndResolve sdpisolver_sdpa.cpp:369-494).  checksdpi test 11
    (min y, y I - [1 2; 2 4] psd, optimum y = 5) with a loose solver tolerance (1e-2) and a tight outer one (1e-6), warm started
    from a point that meets the loose tolerances but lies 2e-3 outside the cone: the engine stops at once ("optimal" by its own
    measure), the backend's check of y finds lambda_min < -feastol, tightens the solver's tolerance by 0.1 and solves again -
    cold, the start point is gone - until y passes."""
    P = sdpi_prepare.prepare(build(case("test11")))
    s = sdpi_call.SdpiSolver(gpu.lib())
    assert s.set_real(3, 1e-2) == sdpi_call.SCIP_OKAY      # SDPSOLVERFEASTOL: what the engine is asked for
    assert s.set_real(1, 1e-2) == sdpi_call.SCIP_OKAY      # GAPTOL
    assert s.set_real(2, 1e-6) == sdpi_call.SCIP_OKAY      # FEASTOL: what y is checked against
    C2 = np.array([[1.0, 2.0], [2.0, 4.0]])
    y0 = 5.0 - 2e-3
    Z0 = 5.0 * np.eye(2) - C2 + 1e-4 * np.eye(2)
    X0 = np.array([[0.2, 0.4], [0.4, 0.8]]) + 1e-4 * np.eye(2)
    tri = lambda M: ([0, 1, 1], [0, 0, 1], [M[0, 0], M[1, 0], M[1, 1]])
    start = dict(y=[y0], Z=[tri(Z0), ([], [], [])], X=[tri(X0), ([], [], [])])
    assert np.linalg.eigvalsh(y0 * np.eye(2) - C2)[0] < -1e-3         # the start y is infeasible by 2e-3
    rc, _, _ = s.solve(P, start=start)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal") and s.flag("IsAcceptable")
    assert s.sdpcalls() >= 2                                           # the first solve was rejected by the check of y
    rc, obj, y = s.dual_sol()
    assert np.linalg.eigvalsh(y[0] * np.eye(2) - C2)[0] >= -1e-6       # what sdpsolchecker.c:201-257 tests
    assert abs(obj - 5.0) <= 1e-2 and s.settings_used() == sdpi_call.FAST
    first_iters = s.iterations()
    # the same call without the start point: one engine solve is enough (its iterates are interior, y + residual stays psd)
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal") and s.sdpcalls() == 1 and s.iterations() <= first_iters
    s.free()


def test_penaltybound_both_ways(gpu):
    """row a6 (sdpisolver_dsdp.c:1655-1734): feasorig <=> r < feastol; for an infeasible node r stays positive and the trace of
    X is pressed against Gamma (penaltybound TRUE: raising Gamma will not help, sdpi.c:3542-3580); for a feasible node
    feasorig is TRUE and penaltybound FALSE.  Same verdicts from the restated backend on the numpy IPM."""
    s = sdpi_call.SdpiSolver(gpu.lib())
    for par in (1, 2, 3):
        assert s.set_real(par, 1e-6) == sdpi_call.SCIP_OKAY
    ref = drv.OracleBackend(feastol=1e-6, gaptol=1e-6)
    for name, gamma in (("test9", 10.0), ("test9", 1e3), ("test10", 1e5)):
        P = sdpi_prepare.prepare(build(case(name)))
        rc, feasorig, pbound = s.solve(P, penaltyparam=gamma, withobj=True, rbound=True)
        assert rc == sdpi_call.SCIP_OKAY and s.flag("IsAcceptable") and s.settings_used() == sdpi_call.PENALTY
        _, feas_ref, pb_ref = ref.solve(P, penaltyparam=gamma, withobj=True, rbound=True)
        assert (feasorig, pbound) == (feas_ref, pb_ref), (name, gamma, feasorig, pbound, feas_ref, pb_ref)
        if name == "test9":                     # infeasible: r > 0, Tr(X) = Gamma
            assert not feasorig and pbound
            _, obj, _ = s.dual_sol()
            assert abs(obj - ref.objval()) <= 1e-5 * max(1.0, abs(obj))          # the solver's own objective (r is part of it)
        else:
            assert feasorig and not pbound
    s.free()


def test_max_primal_entry(gpu):
    """row a13 (sdpisolver_sdpa.cpp:3090-3125): largest entry of the primal matrix over SDP blocks and LP multipliers"""
    s = sdpi_call.SdpiSolver(gpu.lib())
    for par in (1, 2, 3):
        assert s.set_real(par, 1e-6) == sdpi_call.SCIP_OKAY
    assert s.max_primal_entry() == 0.0                                   # nothing solved yet
    rc, _, _ = s.solve(sdpi_prepare.prepare(build(case("test11"))))      # X = [0.2 0.4; 0.4 0.8] (checksdpi.c:1113)
    assert rc == sdpi_call.SCIP_OKAY and abs(s.max_primal_entry() - 0.8) <= 1e-5
    rc, _, _ = s.solve(sdpi_prepare.prepare(build(case("test1"))))       # LP only: multipliers lb (0, 0.5), rhs (1.5, 0) (checksdpi.c:537)
    assert rc == sdpi_call.SCIP_OKAY and abs(s.max_primal_entry() - 1.5) <= 1e-5
    rc, mats = s.primal_solution_matrix()
    s.free()


def test_start_settings_are_honoured(gpu):
    """sdpisolver_sdpa.cpp:1415-1449: a parent node's settings are handed down; SettingsUsed reports the rung that solved"""
    P = sdpi_prepare.prepare(build(case("test10")))
    s = sdpi_call.SdpiSolver(gpu.lib())
    for par in (1, 2, 3):
        assert s.set_real(par, 1e-6) == sdpi_call.SCIP_OKAY
    ref = drv.OracleBackend(feastol=1e-6, gaptol=1e-6, ladder=True)
    objs = []
    for st in (sdpi_call.UNSOLVED, sdpi_call.FAST, sdpi_call.MEDIUM, sdpi_call.STABLE):
        rc, _, _ = s.solve(P, startsettings=st)
        assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
        ref.solve(P, startsettings=st)
        expect = sdpi_call.FAST if st in (sdpi_call.UNSOLVED, sdpi_call.FAST) else st
        assert s.settings_used() == expect == ref.settings_used()
        assert s.sdpcalls() == 1
        assert abs(s.iterations() - ref.iterations()) <= 1            # same rung, same arithmetic: same iteration count
        objs.append(s.objval())
    assert max(objs) - min(objs) <= 1e-5
    assert s.lib.SCIPsdpiSolverLoadAndSolve is not None
    rc, _, _ = s.solve(P, startsettings=7)                             # unknown setting: sdpisolver_sdpa.cpp:1445-1449
    assert rc == sdpi_call.SCIP_LPERROR
    s.free()


def test_two_solver_instances_on_two_host_threads(gpu):
    """sdpisolver.h has no locks: one SCIP_SDPISOLVER is used by one thread at a time, several may be live on different threads
    (concurrent mode) and Free may run on another thread than Create (sdpisolver_mosek.c:472,531-535).  Two threads solve
    different known-answer problems in a loop at the same time; every result must equal the sequential one, and the solvers are
    freed by the main thread."""
    lib = gpu.lib()
    names = ["test11", "test10"]
    seq = {}
    for nm in names:
        s = sdpi_call.SdpiSolver(lib)
        for par in (1, 2, 3):
            s.set_real(par, 1e-6)
        rc, _, _ = s.solve(sdpi_prepare.prepare(build(case(nm))))
        assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
        seq[nm] = (s.dual_sol()[2].copy(), s.iterations())
        s.free()
    solvers = {nm: sdpi_call.SdpiSolver(lib) for nm in names}
    errors = []

    def work(nm):
        try:
            s = solvers[nm]
            for par in (1, 2, 3):
                s.set_real(par, 1e-6)
            P = sdpi_prepare.prepare(build(case(nm)))
            for _ in range(25):
                rc, _, _ = s.solve(P)
                assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
                _, _, y = s.dual_sol()
                assert np.array_equal(y, seq[nm][0]) and s.iterations() == seq[nm][1]
        except Exception as e:                        # pragma: no cover - reported below
            errors.append((nm, repr(e)))

    th = [threading.Thread(target=work, args=(nm,)) for nm in names]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for s in solvers.values():
        s.free()
