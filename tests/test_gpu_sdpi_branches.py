"""Branches of the solver interface that the known-answer tests do not reach (the settings
ladder of the backend, re-entrancy): each one is forced through the public SCIPsdpiSolver* surface of libhipsdp.so."""
import threading
import json
import os
import numpy as np
import pytest

import ipm_ref
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


@pytest.mark.parametrize("seed,expect", [(132, [5, 2, 2]), (163, [5, 2, 2])])
def test_settings_ladder_rescues_what_the_fast_settings_lose(gpu, seed, expect):
    """The rungs of the backend's ladder (hipsdp_params.settings = 0 / 1 / 2; sdpisolver_sdpa.cpp:1698-1795) on two problems of
    the stress family (tests/stress_cases.py) with far more variables than the matrix space has dimensions (seed 132: 5 x 5
    block, 140 variables; seed 163: blocks 2, 2, 5, 64 variables, 3 LP rows): the Schur complement is rank deficient, the fast
    settings end with a numerical failure in the oracle and a more conservative rung finds the ray.  Engine and oracle climb the
    same ladder and end with the same verdict; the rounding of the two implementations may put the rescue of such a problem on
    neighbouring rungs (the engine may already succeed one rung earlier), and the engine's answer is verified on its own."""
    import ipm_ref
    import checker
    import stress_cases
    core, kind = stress_cases.rand_core(np.random.default_rng(seed))
    ref = [ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, settings=lv)) for lv in range(3)]
    assert [r.status for r in ref] == expect
    ref_final = next(r for r in ref if r.status < 4)
    ref_rung = [r.status < 4 for r in ref].index(True)
    s = gpu.Solver(0)
    s.load_core(core)
    got = []
    final = None
    for lv in range(3):
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, settings=lv)
        assert info.settings_used == lv
        got.append(info.status)
        if info.status < 4 and final is None:
            final = (lv, info, s.y(), [s.X(k) for k in range(len(core.blocks))], s.lp()[0])
    s.close()
    assert final is not None, got
    lv, info, y, X, x = final
    assert info.status == ref_final.status, (got, expect)
    assert abs(lv - ref_rung) <= 1, (got, expect)
    assert all(g < 4 for g in got[lv:]), got                    # once a rung has it, the more conservative ones have it too
    if info.status == 0:
        assert abs(info.dobj - ref_final.dobj) <= 1e-5 * (1 + abs(ref_final.dobj))
        ok, det = checker.certificate(core, y, X, x, 1e-5, 1e-5)
        assert ok, det
    elif info.status == 2:                                     # y-ray: b^T y < 0, A^T y psd (checksdpi test 2's kind of verdict)
        assert core.b @ y < 0
        for A in core.blocks:
            Zr = np.tensordot(y, A[1:], axes=(0, 0))
            assert np.linalg.eigvalsh(0.5 * (Zr + Zr.T))[0] >= -1e-6 * abs(core.b @ y)
        if core.q:
            assert np.min(core.D @ y) >= -1e-6 * abs(core.b @ y)


def test_ladder_at_the_interface_settles_the_hard_nodes_of_example_small(gpu):
    """example_small has node relaxations whose optimum is not attained (tau -> 0): before the ladder they went to the caller's
    penalty loop.  Every node through SCIPsdpiSolverLoadAndSolve: optimum -8 (check/testset/short.solu:1), no unresolved node,
    SettingsUsed reports the rung that solved each node; the restated backend on the numpy IPM walks the same tree."""
    import bnb
    import sdpa_io
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", "example_small.dat-s"))
    prob = bnb.instance_to_sdpi(inst)

    def run(be):
        used = []

        def solve(P):
            rc, _, _ = be.solve(P)
            assert rc == sdpi_call.SCIP_OKAY
            used.append(be.settings_used())
            if be.flag("IsDualInfeasible"):
                return bnb.NodeResult('infeasible')
            if be.flag("IsDualUnbounded"):
                return bnb.NodeResult('unbounded')
            if not be.flag("IsOptimal"):
                return bnb.NodeResult('failed')
            rc, obj, y = be.dual_sol()
            return bnb.NodeResult('optimal', obj, y)
        best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, solve)
        return best, nodes, failed, used

    s = sdpi_call.SdpiSolver(gpu.lib())
    for par in (1, 2, 3):
        assert s.set_real(par, 1e-6) == sdpi_call.SCIP_OKAY
    best, nodes, failed, used = run(s)
    s.free()
    rbest, rnodes, rfailed, rused = run(drv.OracleBackend(feastol=1e-6, gaptol=1e-6, ladder=True))
    assert abs(best + 8.0) <= 1e-5 and abs(rbest + 8.0) <= 1e-5
    assert failed == 0 and rfailed == 0
    assert all(u in (sdpi_call.FAST, sdpi_call.MEDIUM, sdpi_call.STABLE) for u in used + rused)
    print("example_small: rungs used by the HIP backend %s, by the numpy backend %s" % (used, rused))
    assert abs(nodes - rnodes) <= 2


def test_node_without_attained_optimum_converges_on_the_fast_settings_like_the_oracle(gpu):
    """The node lb = (-10, -10, -10), ub = (0, 1, 10) of example_small: its optimum -8 is not attained (tau -> 0 with linear
    convergence) and the Schur factor reaches cond(L) = 1e8 on the way.  With the triangular solves as x = inv(L_bb) r the residual
    of M dy = h - the primal infeasibility a step leaves behind - stalled at 1e-6 and the node needed all three rungs of the
    ladder; with every diagonal-block solve corrected once by the factor itself (chol.hip hs_trsv mode bit 4, RB_SOLVE,
    k_solve2_small; DESIGN.md, Robustness) the engine follows the oracle: the fast settings, the same number of iterations, no
    second backend call."""
    import ipm_ref
    import bnb
    import sdpa_io
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", "example_small.dat-s"))
    prob = bnb.instance_to_sdpi(inst)
    node = sdpi_prepare.SdpiProblem(prob.obj, [-10.0, -10.0, -10.0], [0.0, 1.0, 10.0], prob.blocks, prob.lp, isintegral=prob.isintegral)
    P = sdpi_prepare.prepare(node)
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    ref = ipm_ref.hsd_solve(ipm_ref.CoreProblem(b, blk, D, c), ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    assert ref.status == ipm_ref.STATUS_OPTIMAL
    s = sdpi_call.SdpiSolver(gpu.lib())
    for par in (1, 2, 3):
        assert s.set_real(par, 1e-6) == sdpi_call.SCIP_OKAY
    rc, _, _ = s.solve(P)
    assert rc == sdpi_call.SCIP_OKAY and s.flag("IsOptimal")
    assert s.settings_used() == sdpi_call.FAST and s.sdpcalls() == 1
    assert abs(s.iterations() - ref.iterations) <= 1, (s.iterations(), ref.iterations)
    rc, obj, y = s.dual_sol()
    assert abs(obj + 8.0) <= 1e-4
    s.free()


def test_kernel_attributes_are_kept_per_device(gpu):
    """hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair: the engine raises it once per pair (a bit
    per device in a per-kernel mask, csrc/hs_util.cpp) instead of once per process - a second device in the same process gets its
    own.  On a one-GPU box: solver instances pinned with HIPSDP_DEVICE=0 raise the limits of the kernels they launch exactly once
    (the count grows with the first solve and not with later ones or with a second instance), and nothing is booked on device 1."""
    import instances
    lib = gpu.lib()
    b, A, ys, Xs, Zs = instances.planted_dense(150, 200)
    core = ipm_ref.CoreProblem(b, [A])
    os.environ["HIPSDP_DEVICE"] = "0"
    try:
        counts = []
        for _ in range(2):
            s = gpu.Solver(0)
            s.load_core(core)
            for _ in range(2):
                info = s.solve(gaptol=1e-6, feastol=1e-6)
                assert info.status == 0
                counts.append(lib.hipsdp_func_attr_sets(0))
            s.close()
    finally:
        os.environ.pop("HIPSDP_DEVICE", None)
    assert counts[0] > 0
    assert counts[1] == counts[0] and counts[2] == counts[0] and counts[3] == counts[0]
    assert lib.hipsdp_func_attr_sets(1) == 0
