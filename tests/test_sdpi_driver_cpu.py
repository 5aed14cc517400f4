"""CPU tests of the SCIPsdpiSolve restatement (oracle/sdpi_driver.py: one-variable shortcut, Slater check, penalty fallback)
with the numpy backend: pins the harness the gpu-marked twin (tests/test_gpu_sdpi_driver.py) drives libhipsdp.so with."""
import json
import os
import numpy as np
import pytest

import sdpi_prepare
import sdpi_driver as drv
import driver_cases as cases
from conftest import GOLDEN

CASES = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))["cases"]}


def build(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    return sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp)


def test_one_variable_shortcut_reproduces_check1dsdp_test5():
    """unittests/src/check1dsdp.c:341: y = 1.541381 (tolerance 1e-6); the backend is never called"""
    case = CASES["check1dsdp_test5"]
    R = drv.sdpi_solve(None, build(case), feastol=1e-6)
    assert R.solved and R.onevar == 'optimal' and R.nsdpcalls == 0
    assert abs(R.y[0] - case["expect"]["dualsol"][0]) <= 2e-6 and abs(R.objval - case["expect"]["objval"]) <= 2e-6


def test_one_variable_shortcut_declines_and_detects_infeasibility():
    # infinite bound -> declined (solveonevarsdp.c:205-209): checksdpi test11 has a free variable
    assert drv.solve_one_var_sdp(1.0, -drv.INF, drv.INF, 2, [(0, 0, 1.0)], [(0, 0, 1.0)], 1e-6)[0] is None
    # negative objective -> declined (solveonevarsdp.c:212)
    assert drv.solve_one_var_sdp(-1.0, 0.0, 1.0, 2, [], [(0, 0, 1.0)], 1e-6)[0] is None
    # y * diag(1, -1) - I >= 0 is infeasible on [0, 3]
    obj, opt, g = drv.solve_one_var_sdp(1.0, 0.0, 3.0, 2, [(0, 0, 1.0), (1, 1, 1.0)], [(0, 0, 1.0), (1, 1, -1.0)], 1e-6)
    assert obj >= drv.INF
    # y * I - diag(1, 2) >= 0 on [0, 5]: optimum y = 2 (within feastol / 2 below)
    obj, opt, g = drv.solve_one_var_sdp(3.0, 0.0, 5.0, 2, [(0, 0, 1.0), (1, 1, 2.0)], [(0, 0, 1.0), (1, 1, 1.0)], 1e-6)
    assert abs(opt - 2.0) <= 1e-6 and abs(obj - 6.0) <= 3e-6
    # lower bound already feasible
    obj, opt, g = drv.solve_one_var_sdp(1.0, 4.0, 5.0, 2, [(0, 0, 1.0)], [(0, 0, 1.0), (1, 1, 1.0)], 1e-6)
    assert opt == 4.0 and obj == 4.0


def test_regular_node_goes_through_one_backend_call():
    prob, Cm = cases.maxcut_like()
    be = drv.OracleBackend()
    R = drv.sdpi_solve(be, prob)
    assert R.solved and not R.penalty and not R.infeasible and R.nsdpcalls == 1
    Z = np.diag(R.y) - Cm
    assert np.linalg.eigvalsh(Z)[0] >= -1e-5
    # forcing the fallback on the same problem walks feasibility problem + penalty loop and ends at the same optimum
    R2 = drv.sdpi_solve(drv.OracleBackend(), prob, force_penalty=True)
    assert R2.solved and R2.penalty and not R2.infeasible and R2.npenaltysolves == 2 and R2.penaltyparam_used == drv.DEFAULT_PENALTYPARAM
    assert abs(R2.objval - R.objval) <= 1e-4 * max(1.0, abs(R.objval))


def test_penalty_fallback_proves_infeasibility():
    R = drv.sdpi_solve(drv.OracleBackend(), cases.infeasible_block(), force_penalty=True)
    assert R.penalty and R.infeasible and R.npenaltysolves == 1
    # the regular call finds it as well (Farkas ray)
    R = drv.sdpi_solve(drv.OracleBackend(), cases.infeasible_block())
    assert R.infeasible and not R.penalty


def test_slater_check():
    prob, _ = cases.maxcut_like()
    R = drv.sdpi_solve(drv.OracleBackend(), prob, slatercheck=True)
    assert R.dualslater == drv.SLATER_HOLDS and R.primalslater == drv.SLATER_HOLDS        # all variables bounded (sdpi.c:1771)
    R = drv.sdpi_solve(drv.OracleBackend(), cases.no_interior(), slatercheck=True)
    assert R.dualslater == drv.SLATER_NOT
    assert R.solved and abs(R.objval - 1.0) <= 1e-4
    R = drv.sdpi_solve(drv.OracleBackend(), cases.infeasible_block(), slatercheck=True)
    assert R.dualslater == drv.SLATER_INF and R.infeasible
    prob = cases.free_variable_with_rows()
    R = drv.sdpi_solve(drv.OracleBackend(), prob, slatercheck=True)
    assert R.dualslater == drv.SLATER_HOLDS and R.primalslater in (drv.SLATER_HOLDS, drv.SLATER_NOT)
    assert R.solved and abs(R.objval - 1.0) <= 1e-4 and abs(R.y[1]) <= 1e-2


def test_primal_slater_arguments_follow_the_reference_layout():
    prob = cases.free_variable_with_rows()
    P = sdpi_prepare.prepare(prob)
    Q, allbounded = drv.primal_slater_arguments(P)
    assert not allbounded
    assert Q.nlpcons == P.nlpcons + 1 and Q.lpnnonz == P.lpnnonz + 1        # tr(A_0) = 2, tr(A_1) = 0 -> one entry
    assert Q.lpbeg[P.nlpcons] == P.lpnnonz and Q.lpind[P.lpnnonz] == 0 and Q.lpval[P.lpnnonz] == 2.0
    assert Q.lplhs[P.nlpcons] == 1.0 and Q.lprhs[P.nlpcons] >= drv.INF
    assert Q.lplhs[0] == 0.0 and Q.lprhs[0] >= drv.INF                      # finite side -> 0, infinite side kept
    assert list(Q.lb) == [0.0, -drv.INF] and list(Q.ub) == [0.0, drv.INF]  # finite bounds -> 0 (variable 0 becomes fixed)
    assert all(len(c) == 0 for c in Q.sdpconst)
