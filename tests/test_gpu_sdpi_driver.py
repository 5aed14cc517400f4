"""GPU: the restated SCIPsdpiSolve driver (oracle/sdpi_driver.py - one-variable shortcut, Slater check, penalty fallback,
sdpi.c:3123-3640) on top of libhipsdp.so: SCIPsdpiSolverLoadAndSolve[WithPenalty] receive exactly the call sequences and
argument patterns the reference's SDPI produces around a node solve (feasibility problem with free r, penalty solves with
Gamma = 1e5.., Slater problems with zeroed bounds and an appended row), and the eigenvalue problems of the one-variable
shortcut go through SCIPlapackComputeIthEigenvalue of the same library."""
import ctypes as C
import json
import os
import numpy as np
import pytest

import bnb
import sdpa_io
import sdpi_prepare
import sdpi_call
import sdpi_driver as drv
import driver_cases as cases
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))["cases"]}
PD = C.POINTER(C.c_double)


def build(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    return sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp)


def backend(gpu, tol=1e-6):
    s = sdpi_call.SdpiSolver(gpu.lib())
    for par in (1, 2, 3):                       # GAPTOL, FEASTOL, SDPSOLVERFEASTOL
        assert s.set_real(par, tol) == sdpi_call.SCIP_OKAY
    return s


def hip_lmin(gpu):
    lib = gpu.lib()

    def lmin(M):
        n = M.shape[0]
        a = np.ascontiguousarray(M, dtype=np.float64).copy()      # the routine destroys its input (lapack_interface.c:178-288)
        val = C.c_double(0.0)
        vec = np.zeros(n)
        assert lib.SCIPlapackComputeIthEigenvalue(None, 1, n, a.ctypes.data_as(PD), 1, C.byref(val), vec.ctypes.data_as(PD)) == 1
        return val.value, vec
    return lmin


def test_one_variable_shortcut_with_device_eigenvalues(gpu):
    case = CASES["check1dsdp_test5"]                              # unittests/src/check1dsdp.c:341
    R = drv.sdpi_solve(None, build(case), feastol=1e-6, lmin=hip_lmin(gpu))
    assert R.solved and R.onevar == 'optimal'
    assert abs(R.y[0] - case["expect"]["dualsol"][0]) <= 2e-6 and abs(R.objval - case["expect"]["objval"]) <= 2e-6
    # the same problem through the backend instead of the shortcut
    be = backend(gpu)
    R2 = drv.sdpi_solve(be, build(case), onevar_shortcut=False)
    be.free()
    assert R2.solved and R2.onevar is None and abs(R2.objval - case["expect"]["objval"]) <= 2e-6


def test_regular_node_and_forced_penalty_fallback(gpu):
    prob, Cm = cases.maxcut_like()
    ref = drv.sdpi_solve(drv.OracleBackend(), prob)
    be = backend(gpu)
    R = drv.sdpi_solve(be, prob)
    assert R.solved and not R.penalty and R.nsdpcalls == 1
    assert abs(R.objval - ref.objval) <= 1e-5 and np.max(np.abs(R.y - ref.y)) <= 1e-4
    R2 = drv.sdpi_solve(be, prob, force_penalty=True)
    be.free()
    assert R2.solved and R2.penalty and not R2.infeasible and R2.npenaltysolves == 2
    assert abs(R2.objval - ref.objval) <= 1e-4 * max(1.0, abs(ref.objval))
    assert np.linalg.eigvalsh(np.diag(R2.y) - Cm)[0] >= -1e-5


def test_penalty_fallback_proves_infeasibility(gpu):
    be = backend(gpu)
    R = drv.sdpi_solve(be, cases.infeasible_block(), force_penalty=True)
    assert R.penalty and R.infeasible and R.npenaltysolves == 1
    R = drv.sdpi_solve(be, cases.infeasible_block())
    be.free()
    assert R.infeasible and not R.penalty


def test_slater_check_matches_the_numpy_backend(gpu):
    be = backend(gpu)
    for make in (lambda: cases.maxcut_like()[0], cases.no_interior, cases.infeasible_block, cases.free_variable_with_rows):
        ref = drv.sdpi_solve(drv.OracleBackend(), make(), slatercheck=True)
        R = drv.sdpi_solve(be, make(), slatercheck=True)
        assert (R.dualslater, R.primalslater) == (ref.dualslater, ref.primalslater), make
        assert R.infeasible == ref.infeasible and R.solved == ref.solved
        if ref.objval is not None:
            assert abs(R.objval - ref.objval) <= 1e-4
    be.free()


@pytest.mark.parametrize("name,optimum", [("example_small.dat-s", -8.0), ("example_tightenmatrices.dat-s", -9.0),
                                          ("example_TT.dat-s.gz", 2.11803), ("example_CLS.dat-s.gz", 7.1485), ("example_inf.dat-s", None),
                                          ("example_MkP.dat-s.gz", -95.0)])
def test_bnb_with_the_full_driver_per_node(gpu, name, optimum):
    """every node goes through the driver: one-variable nodes are decided by the shortcut (device eigenvalues), nodes the
    backend cannot solve acceptably fall back to the penalty formulation like sdpi.c:3437"""
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", name))
    prob = bnb.instance_to_sdpi(inst, integrality=True)      # as SCIP-SDP runs: LP rows tightened by integrality at every node
    be = backend(gpu)
    lmin = hip_lmin(gpu)
    stats = dict(onevar=0, penalty=0, calls=0)

    def solve(P):
        R = drv.sdpi_solve(be, P.prob, prepared=P, lmin=lmin, feastol=1e-6, gaptol=1e-6)
        stats["calls"] += R.nsdpcalls
        stats["onevar"] += 1 if R.onevar else 0
        stats["penalty"] += 1 if R.penalty else 0
        if R.infeasible:
            return bnb.NodeResult('infeasible')
        if not R.solved or R.objval is None:
            return bnb.NodeResult('failed')
        return bnb.NodeResult('optimal', R.objval, R.y)

    best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, solve)
    be.free()
    print("%s through the driver: optimum %s, %d nodes, %d backend calls, %d one-variable nodes, %d penalty nodes, %d unresolved" %
          (name, best, nodes, stats["calls"], stats["onevar"], stats["penalty"], failed))
    if optimum is None:
        assert best is None
        return
    assert best is not None and abs(best - optimum) <= 1e-4 * max(1.0, abs(optimum))
    assert all(abs(y[v] - round(y[v])) <= 1e-9 for v in inst.intvars)
    # measured: 0 unresolved nodes on all six instances.  example_MkP (check/testset/short.solu:7; 105 binaries, one 15 x 15
    # block, 240 LP rows) has about fifty nodes without interior that go through the penalty formulation (56 of 101 nodes; the
    # numpy backend behind the same driver leaves 7 of its 119 nodes unresolved); a node left unresolved is simply branched on
    # (round 3: with the noise-level pivot rule of the semidefinite Cholesky no node of example_MkP is left unresolved either)
    assert failed == 0
