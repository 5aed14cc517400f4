"""GPU parity tests of the whole node solve (engine level, C ABI hipsdp_*): same inputs through the HIP engine and the
oracle, plus algorithm-independent certificates, plus size-independent properties at BASELINE's full size."""
import json
import os
import numpy as np
import pytest

import ipm_ref
import checker
import instances
import sdpa_io
import sdpi_prepare
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-5          # north_star: results agree to sdpsolvergaptol / feastol = 1e-5 (relax_sdp.c:70-71)


def gpu_solve(hb, core, **kw):
    s = hb.Solver(0)
    s.load_core(core)
    info = s.solve(**kw)
    out = dict(info=info, y=s.y(), X=[s.X(k) for k in range(len(core.blocks))], lp=s.lp())
    s.close()
    return out


def compare(hb, core, tol=1e-6):
    # solver tolerance tol, acceptance (TOL = 10 tol) absolute on the X-side: the reference's sdpsolverfeastol / feastol pair
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=tol, feastol=tol, pabstol=10 * tol))
    g = gpu_solve(hb, core, gaptol=tol, feastol=tol, pabstol=10 * tol)
    assert g["info"].status == ref.status, (g["info"].status, ref.status)
    return ref, g


CASES = json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))["cases"]


def case_core(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    P = sdpi_prepare.prepare(sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp))
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    return ipm_ref.CoreProblem(b, blk, D, c)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_engine_matches_oracle_on_reference_cases(gpu, case):
    core = case_core(case)
    ref, g = compare(gpu, core)
    assert g["info"].iterations == ref.iterations
    # test4 is infeasible on both sides with a singular Schur complement ([1 -1; -1 1]): its second pivot is zero in exact
    # arithmetic, the sign of the rounding residue decides whether the pivot rule zeroes the column, and y along the null
    # direction follows that coin: the verdict and the iteration count are compared, y only to three digits
    ytol = 1e-3 if (case["expect"].get("primal"), case["expect"].get("dual")) == ("infeas", "infeas") else 1e-7
    assert np.allclose(g["y"], ref.y, atol=ytol, rtol=ytol)
    if core.q:
        assert np.allclose(g["lp"][0], ref.x, atol=1e-6, rtol=1e-6)
    for Xg, Xr in zip(g["X"], ref.X):
        assert np.allclose(Xg, Xr, atol=1e-6, rtol=1e-6)
    if ref.status == ipm_ref.STATUS_OPTIMAL:
        ok, det = checker.certificate(core, g["y"], g["X"], g["lp"][0], TOL, TOL)
        assert ok, det
    if ref.status in (ipm_ref.STATUS_DINF, ipm_ref.STATUS_PDINF):
        assert checker.farkas_dual_infeasible(core, g["X"], g["lp"][0], 1e-6)[0]
    if ref.status in (ipm_ref.STATUS_DUNB, ipm_ref.STATUS_PDINF):
        assert checker.farkas_dual_unbounded(core, g["y"], 1e-6)[0]


@pytest.mark.parametrize("name", ["example_small.dat-s", "example_TT.dat-s.gz", "example_CLS.dat-s.gz", "example_MkP.dat-s.gz",
                                  "example_inf.dat-s", "example_tightenmatrices.dat-s"])
def test_engine_matches_oracle_on_reference_instances(gpu, name):
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", name))
    D, c = sdpa_io.lp_dense(inst)
    core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
    ref, g = compare(gpu, core)
    assert abs(g["info"].dobj - ref.dobj) <= TOL * (1 + abs(ref.dobj))
    ok, det = checker.certificate(core, g["y"], g["X"], g["lp"][0], TOL, TOL)
    assert ok, det


@pytest.mark.parametrize("n,m", [(5, 8), (20, 40), (50, 100), (100, 200)])
def test_engine_matches_oracle_on_planted_blocks(gpu, n, m):
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    core = ipm_ref.CoreProblem(b, [A])
    ref, g = compare(gpu, core, tol=1e-5)
    assert g["info"].iterations == ref.iterations
    assert abs(g["info"].dobj - ref.dobj) <= 1e-8 * (1 + abs(ref.dobj))
    assert np.max(np.abs(g["y"] - ref.y)) <= 1e-6          # well inside the 1e-5 parity tolerance; stops at the same iterate
    assert abs(g["info"].dobj - b @ ys) <= TOL * (1 + abs(b @ ys))


@pytest.mark.parametrize("n,m,two_blocks", [(100, 200, False), (150, 320, False), (96, 260, True)])
def test_first_assembly_of_a_cold_solve_is_the_gram_matrix_of_the_constraint_matrices(gpu, monkeypatch, n, m, two_blocks):
    """At the cold start X = Z = xi I the Schur complement is M_ij = tr(A_i X A_j Z^-1) = <A_i, A_j>: the engine's first assembly runs the
    Gram product on the rows of A alone (csrc/schur.hip: hs_schur_W_identity) instead of multiplying by sqrt(xi) I and I / sqrt(xi)
    first.  With HIPSDP_NO_IDENTITY_START=1 the same solve takes the general products in that iteration too: the same iteration
    count, the same objective and y to rounding - and both agree with the oracle, which knows nothing of the shortcut."""
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    blocks = [A]
    if two_blocks:
        # a second dense block that keeps the problem strictly feasible on both sides: Z_2(y*) = Zs2 > 0, X_2 = I
        rng = np.random.default_rng(11)
        n2 = 70
        A2 = rng.standard_normal((m + 1, n2, n2)) / np.sqrt(2.0 * n2)
        A2 = A2 + A2.transpose(0, 2, 1)
        Zs2 = rng.standard_normal((n2, n2)); Zs2 = Zs2 @ Zs2.T / n2 + 0.5 * np.eye(n2)
        A2[0] = np.tensordot(ys, A2[1:], axes=(0, 0)) - Zs2
        blocks.append(A2)
        b = b + np.array([np.trace(A2[i]) for i in range(1, m + 1)])
    core = ipm_ref.CoreProblem(b, blocks)
    monkeypatch.delenv("HIPSDP_NO_IDENTITY_START", raising=False)
    g1 = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.setenv("HIPSDP_NO_IDENTITY_START", "1")
    g2 = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.delenv("HIPSDP_NO_IDENTITY_START", raising=False)
    assert g1["info"].status == 0 and g2["info"].status == 0
    assert g1["info"].iterations == g2["info"].iterations
    assert abs(g1["info"].dobj - g2["info"].dobj) <= 1e-11 * (1 + abs(g2["info"].dobj))
    assert np.max(np.abs(g1["y"] - g2["y"])) <= 1e-9 * (1 + np.max(np.abs(g2["y"])))
    # the first assembly executes the Gram product only: fewer matrix-core flops than with the two n^3 products
    assert g1["info"].schur_flops_executed < g2["info"].schur_flops_executed
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    assert ref.status == 0 and ref.iterations == g1["info"].iterations
    assert np.max(np.abs(g1["y"] - ref.y)) <= 1e-6


def test_multi_block_with_lp_rows_and_bounds(gpu):
    rng = np.random.default_rng(7)
    m = 12
    blocks = []
    for n in (4, 9):
        A = rng.standard_normal((m + 1, n, n))
        A = A + A.transpose(0, 2, 1)
        A[0] = -np.eye(n) * 3.0 + 0.1 * A[0]
        blocks.append(A)
    D = np.concatenate([rng.standard_normal((5, m)), np.eye(m), -np.eye(m)])
    c = np.concatenate([-np.ones(5) * 4.0, -2.0 * np.ones(m), -2.0 * np.ones(m)])
    core = ipm_ref.CoreProblem(rng.standard_normal(m), blocks, D, c)
    ref, g = compare(gpu, core)
    assert ref.status == ipm_ref.STATUS_OPTIMAL
    assert np.max(np.abs(g["y"] - ref.y)) <= 1e-6
    ok, det = checker.certificate(core, g["y"], g["X"], g["lp"][0], TOL, TOL)
    assert ok, det


def test_blocks_of_several_sizes_above_64_take_turns_on_the_step_length_kernel(gpu):
    """Blocks with more than 64 rows get their step lengths from the one-launch Lanczos kernel, whose workgroups hand the
    product vector over through exchange vectors that all blocks of a solver share (eig.hip): a longer block must not find
    entries a shorter one left behind.  Three sizes in the order long - short - middle, against the oracle."""
    rng = np.random.default_rng(11)
    m = 40
    blocks = []
    for n in (150, 70, 101):
        A = rng.standard_normal((m + 1, n, n)) / np.sqrt(n)
        A = A + A.transpose(0, 2, 1)
        A[0] = -np.eye(n) * 3.0 + 0.1 * A[0]
        blocks.append(A)
    D = np.concatenate([np.eye(m), -np.eye(m)])
    c = -2.0 * np.ones(2 * m)
    core = ipm_ref.CoreProblem(rng.standard_normal(m), blocks, D, c)
    ref, g = compare(gpu, core)
    assert ref.status == ipm_ref.STATUS_OPTIMAL
    assert g["info"].iterations == ref.iterations
    assert np.max(np.abs(g["y"] - ref.y)) <= 1e-6
    ok, det = checker.certificate(core, g["y"], g["X"], g["lp"][0], TOL, TOL)
    assert ok, det


def test_device_generator_matches_numpy_stream(gpu):
    n, m = 16, 24
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    s = gpu.Solver(0)
    s.set_shape(m, [n], 0)
    bdev = s.gen_planted(n, m, 20240, Xs, Zs, ys)
    Adev = s.get_block_dense(0)
    s.close()
    assert np.max(np.abs(Adev[1:] - A[1:])) <= 1e-13          # same uniforms; log/cos/sqrt differ by ulps at most
    assert np.max(np.abs(Adev[0] - A[0])) <= 1e-12
    assert np.max(np.abs(bdev - b)) <= 1e-12


def test_full_size_c2_properties(gpu):
    """BASELINE config 2 (n = 500, m = 1000, dense, fp64): the oracle would need minutes here, so the check is by
    size-independent properties: the planted optimum value, the y-side acceptance test of sdpsolchecker.c evaluated with
    LAPACK on the host, complementarity, and run-to-run bitwise reproducibility."""
    n, m = 500, 1000
    Q, _ = np.linalg.qr(instances.counter_normal(20240 + 1000003, np.arange(n * n, dtype=np.uint64)).reshape(n, n))
    r = n // 4
    ev = 1.0 + instances.counter_uniform(20240 + 2000003, np.arange(n, dtype=np.uint64))
    Xs = (Q * np.where(np.arange(n) < r, ev, 0.0)) @ Q.T
    Zs = (Q * np.where(np.arange(n) < r, 0.0, ev)) @ Q.T
    ys = 2.0 * instances.counter_uniform(20240 + 3000003, np.arange(m, dtype=np.uint64)) - 1.0
    s = gpu.Solver(0)
    s.set_shape(m, [n], 0)
    b = s.gen_planted(n, m, 20240, Xs, Zs, ys)
    info = s.solve(gaptol=1e-5, feastol=1e-5)
    y1, X1 = s.y(), s.X(0)
    assert info.status == 0
    opt = float(b @ ys)
    assert abs(info.dobj - opt) <= TOL * (1 + abs(opt))
    assert abs(info.pobj - opt) <= TOL * (1 + abs(opt))
    A = s.get_block_dense(0)
    Z = np.tensordot(y1, A[1:], axes=(0, 0)) - A[0]
    assert np.linalg.eigvalsh(0.5 * (Z + Z.T))[0] >= -TOL                       # sdpsolchecker.c:201-257
    assert np.max(np.abs(A[1:].reshape(m, -1) @ X1.reshape(-1) - b)) <= TOL * (1 + np.abs(b).max())
    assert np.linalg.eigvalsh(X1)[0] >= -TOL
    assert abs(np.sum(X1 * Z)) <= 10 * TOL * (1 + abs(opt))                     # complementarity
    info2 = s.solve(gaptol=1e-5, feastol=1e-5)
    assert np.array_equal(s.y(), y1) and info2.iterations == info.iterations    # deterministic reductions
    lmin, viol = s.check_y(y1)
    assert lmin[0] >= -TOL
    # the check the backend's re-solve loop makes (hipsdp_check_y_tol): a feasible y is certified by one Cholesky factorization of
    # Z(y) + 0.999 tol I and reports that bound; a y that violates the tolerance gets the exact eigenvalue
    lmin_t, _ = s.check_y(y1, tol=1e-5)
    assert abs(lmin_t[0] + 0.999e-5) <= 1e-18
    ybad = y1.copy()
    ybad[0] += 0.05
    lam_exact = np.linalg.eigvalsh(0.5 * ((Z + 0.05 * A[1]) + (Z + 0.05 * A[1]).T))[0]
    assert lam_exact < -1e-3
    lmin_b, _ = s.check_y(ybad, tol=1e-5)
    assert abs(lmin_b[0] - lam_exact) <= 1e-9 * (1 + abs(lam_exact))
    s.close()


@pytest.mark.parametrize("mode,n,m", [("U", 40, 70), ("R", 40, 70), ("K1", 40, 70), ("K2", 40, 70), ("K3", 70, 30), ("K8", 150, 40)])
def test_alternative_schur_formulations_give_the_same_solve(gpu, mode, n, m, monkeypatch):
    """HIPSDP_SCHUR=U: chunked U_j = X A_j Z^-1 formulation; =R: the row-sharded form (all chunks on one device); =K<g>: the
    column-sliced W formulation the ranks of a multi-GPU run execute (all g slices on one device, added up in place of the
    all-reduce; K8 at n = 150 has uneven and n = 40 at K3 would have empty slices).  All must reproduce the default solve and
    the oracle."""
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    core = ipm_ref.CoreProblem(b, [A])
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    base = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.setenv("HIPSDP_SCHUR", mode)
    alt = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.delenv("HIPSDP_SCHUR")
    assert alt["info"].status == 0 and alt["info"].iterations == base["info"].iterations == ref.iterations
    assert np.max(np.abs(alt["y"] - base["y"])) <= 1e-8
    assert np.max(np.abs(alt["y"] - ref.y)) <= 1e-6


def test_rccl_code_path_with_a_single_rank_communicator(gpu):
    """the multi-GPU branch of the engine (row-sharded Schur, ncclAllGather x2, ncclBroadcast of the decision scalars) driven
    through a real RCCL communicator of size 1: every collective call, its buffers and its stream ordering are executed on the
    device; only the inter-GPU transport is not (that needs the multi-GPU bench)."""
    import ctypes as C
    lib = gpu.lib()
    uid = (C.c_ubyte * 128)()
    assert lib.hipsdp_comm_unique_id(uid) == 0
    b, A, ys, Xs, Zs = instances.planted_dense(40, 70)
    core = ipm_ref.CoreProblem(b, [A])
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    s = gpu.Solver(0)
    s.load_core(core)
    comm = C.c_void_p()
    assert lib.hipsdp_comm_create(uid, 0, 1, C.byref(comm)) == 0
    assert lib.hipsdp_set_comm(s.h, comm, 0, 1) == 0
    info = s.solve(gaptol=1e-6, feastol=1e-6)
    y = s.y()
    assert lib.hipsdp_set_comm(s.h, None, 0, 1) == 0
    s.close()
    lib.hipsdp_comm_destroy(comm)
    assert info.status == 0 and info.iterations == ref.iterations
    assert np.max(np.abs(y - ref.y)) <= 1e-6


def test_rccl_all_to_all_and_sharded_passes_with_a_single_rank_communicator(gpu, monkeypatch):
    """the variable-sharded branch (hipsdp_shard_matrices: ncclSend / ncclRecv group of the all-to-all, ncclAllReduce of the partial
    Schur matrix, ncclAllGather / ncclAllReduce of the row-swept passes) through a real RCCL communicator of size 1, two column
    slices per assembly"""
    import ctypes as C
    monkeypatch.setenv("HIPSDP_VAR_SLICE", "48")
    lib = gpu.lib()
    uid = (C.c_ubyte * 128)()
    assert lib.hipsdp_comm_unique_id(uid) == 0
    b, A, ys, Xs, Zs = instances.planted_dense(80, 70)
    core = ipm_ref.CoreProblem(b, [A])
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    s = gpu.Solver(0)
    comm = C.c_void_p()
    assert lib.hipsdp_comm_create(uid, 0, 1, C.byref(comm)) == 0
    assert lib.hipsdp_set_comm(s.h, comm, 0, 1) == 0
    assert lib.hipsdp_shard_matrices(s.h, 1) == 0
    s.load_core(core)
    assert lib.hipsdp_matrices_sharded(s.h) == 1
    info = s.solve(gaptol=1e-6, feastol=1e-6)
    y = s.y()
    assert lib.hipsdp_set_comm(s.h, None, 0, 1) == 0
    s.close()
    lib.hipsdp_comm_destroy(comm)
    assert info.status == 0 and abs(info.iterations - ref.iterations) <= 1
    assert np.max(np.abs(y - ref.y)) <= 1e-6


def _solve_with_start(hb, core, start, tol=1e-6):
    s = hb.Solver(0)
    s.load_core(core)
    if start is not None:
        s.set_start(*start)
    info = s.solve(gaptol=tol, feastol=tol)
    out = dict(info=info, y=s.y(), X=[s.X(k) for k in range(len(core.blocks))], Z=[s.Z(k) for k in range(len(core.blocks))], lp=s.lp())
    s.close()
    return out


@pytest.mark.parametrize("n,m", [(30, 45), (130, 160)])
def test_warm_start_from_an_interior_point_matches_oracle_and_saves_iterations(gpu, n, m):
    """(f)-3: a start point handed in through hipsdp_set_start (what SCIPsdpiSolverLoadAndSolve receives as starty / startZ /
    startX, sdpisolver.h:160-173) is used when strictly interior; the oracle applies the same rule."""
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    core = ipm_ref.CoreProblem(b, [A])
    cold = _solve_with_start(gpu, core, None)
    assert cold["info"].status == 0 and cold["info"].warm_started == 0
    # the optimum pushed into the interior (relax_sdp.c builds its warm-start points the same way: convex combination with
    # a scaled identity)
    d = 1e-2
    y0 = cold["y"]
    X0 = [(1 - d) * cold["X"][0] + d * np.eye(n)]
    Z0 = [(1 - d) * cold["Z"][0] + d * np.eye(n)]
    warm = _solve_with_start(gpu, core, (y0, X0, Z0))
    st = ipm_ref.warm_start_point(core, y0, X0, Z0, np.zeros(0), np.zeros(0))
    assert st is not None
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6), start=st)
    assert warm["info"].status == 0 and warm["info"].warm_started == 1
    assert warm["info"].iterations == ref.iterations
    assert warm["info"].iterations < cold["info"].iterations
    assert abs(warm["info"].dobj - cold["info"].dobj) <= 1e-6 * (1 + abs(cold["info"].dobj))
    assert np.max(np.abs(warm["y"] - ref.y)) <= 1e-6


def test_warm_start_that_is_not_interior_falls_back_to_the_cold_start(gpu):
    b, A, ys, Xs, Zs = instances.planted_dense(20, 30)
    core = ipm_ref.CoreProblem(b, [A])
    cold = _solve_with_start(gpu, core, None)
    # X* and Z* themselves are singular (complementary): no Cholesky factor, the engine must ignore the point
    bad = _solve_with_start(gpu, core, (ys, [Xs], [Zs]))
    assert ipm_ref.warm_start_point(core, ys, [Xs], [Zs], np.zeros(0), np.zeros(0)) is None
    assert bad["info"].warm_started == 0
    assert bad["info"].status == 0 and bad["info"].iterations == cold["info"].iterations
    assert np.max(np.abs(bad["y"] - cold["y"])) <= 1e-12


def test_warm_start_with_lp_rows(gpu):
    case = next(c for c in CASES if c["name"] == "test10")
    core = case_core(case)
    cold = _solve_with_start(gpu, core, None)
    assert cold["info"].status == 0
    x, z = cold["lp"]
    d = 1e-2
    n = core.blocks[0].shape[1]
    start = (cold["y"], [(1 - d) * cold["X"][0] + d * np.eye(n)], [(1 - d) * cold["Z"][0] + d * np.eye(n)], (1 - d) * x + d, (1 - d) * z + d)
    warm = _solve_with_start(gpu, core, start)
    st = ipm_ref.warm_start_point(core, *start)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6), start=st)
    assert warm["info"].warm_started == 1 and warm["info"].status == ref.status == 0
    assert warm["info"].iterations == ref.iterations
    assert np.max(np.abs(warm["y"] - ref.y)) <= 1e-6
    # non-positive multiplier: rejected
    xb = x.copy(); xb[0] = 0.0
    bad = _solve_with_start(gpu, core, (start[0], start[1], start[2], xb, start[4]))
    assert bad["info"].warm_started == 0 and bad["info"].status == 0


def test_eigenvector_cuts_on_the_device(gpu):
    """LP-based mode of the reference (cons_sdp.c:896-1010, :1612-1803): at a point y with Z(y) not psd the device returns,
    for every eigenvalue <= -tol, the cut sum_i (v^T A_i v) y_i >= v^T A_0 v.  Checked against the numpy restatement
    (oracle/eigcuts_ref.py): eigenvalues, coefficients up to the sign of v, violation = -eigenvalue, validity at the planted
    feasible point."""
    import eigcuts_ref
    for (n, m, seed) in [(12, 7, 0), (40, 25, 1), (90, 30, 2), (200, 60, 3)]:
        b, A, ys, Xs, Zs = instances.planted_dense(n, m)
        core = ipm_ref.CoreProblem(b, [A])
        rng = np.random.default_rng(seed)
        y = ys + 0.7 * rng.standard_normal(m)                      # leaves the feasible set
        s = gpu.Solver(0)
        s.load_core(core)
        ev, co, lh, ve = s.eigencuts(0, y, 1e-6, 5)
        rev, rco, rlh, rve = eigcuts_ref.cuts_dense(A, y, 1e-6, 5)
        assert len(ev) == len(rev) and len(ev) >= 1
        assert np.max(np.abs(ev - rev)) <= 1e-9 * max(1.0, np.max(np.abs(rev)))
        for c in range(len(ev)):
            v = ve[c]
            assert abs(np.linalg.norm(v) - 1.0) <= 1e-10
            assert abs(abs(v @ rve[c]) - 1.0) <= 1e-6               # same eigenvector up to sign (eigenvalues are simple here)
            assert np.max(np.abs(co[c] - rco[c])) <= 1e-6 * max(1.0, np.max(np.abs(rco[c])))
            assert abs(lh[c] - rlh[c]) <= 1e-6 * max(1.0, abs(rlh[c]))
            assert abs((co[c] @ y - lh[c]) - ev[c]) <= 1e-8 * max(1.0, abs(ev[c]))     # violated by exactly -lambda
            assert co[c] @ ys - lh[c] >= -1e-9                       # valid for the feasible point
        # nothing to cut at a feasible point; maxcuts = 0 is allowed
        ev0, _, _, _ = s.eigencuts(0, ys, 1e-6, 5)
        assert len(ev0) == 0
        assert len(s.eigencuts(0, y, 1e-6, 0)[0]) == 0
        s.close()


def test_preoptimal_iterate_matches_the_oracle(gpu):
    """params.preoptgap (SCIP_SDPPAR_WARMSTARTPOGAP): the engine keeps the first iterate that is feasible to tolerance with a
    relative gap below it - the same iterate the oracle keeps, strictly interior, and not the final one"""
    rng = np.random.default_rng(8)
    for (n, m, q) in [(12, 9, 0), (30, 40, 6)]:
        b, A, ys, Xs, Zs = instances.planted_dense(n, m)
        if q:
            D = rng.standard_normal((q, m))
            c = D @ ys - rng.uniform(0.1, 1.0, q)
            core = ipm_ref.CoreProblem(b, [A], D, c)
        else:
            core = ipm_ref.CoreProblem(b, [A])
        ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-7, feastol=1e-6, preoptgap=1e-2))
        assert ref.status == 0 and ref.pre is not None and ref.pre["it"] < ref.iterations
        s = gpu.Solver(0)
        s.load_core(core)
        info = s.solve(gaptol=1e-7, feastol=1e-6, preoptgap=1e-2)
        pre = s.preoptimal()
        yfin = s.y()
        assert info.status == 0 and info.iterations == ref.iterations and pre is not None
        y, X, x = pre
        assert np.max(np.abs(y - ref.pre["y"])) <= 1e-6 * max(1.0, np.max(np.abs(ref.pre["y"])))
        assert np.max(np.abs(X[0] - ref.pre["X"][0])) <= 1e-6 * max(1.0, np.max(np.abs(ref.pre["X"][0])))
        if q:
            assert np.max(np.abs(x - ref.pre["x"])) <= 1e-6 * max(1.0, np.max(np.abs(ref.pre["x"])))
        assert np.linalg.eigvalsh(X[0])[0] > 0.0                          # interior
        Z = np.tensordot(y, A[1:], axes=(0, 0)) - A[0]
        assert np.linalg.eigvalsh(Z)[0] > -1e-5
        assert np.max(np.abs(y - yfin)) > 1e-9                            # not the final iterate
        # without the parameter nothing is kept
        s.solve(gaptol=1e-7, feastol=1e-6)
        assert s.preoptimal() is None
        s.close()


@pytest.mark.parametrize("n,m", [(12, 30), (20, 64), (16, 100), (24, 128)])
def test_solves_with_m_by_substitution_in_the_oracles_order(gpu, n, m, monkeypatch):
    """HIPSDP_SMALL_SOLVE=subst (round 6): on the general path the solves with the factor of M for m <= 128 substitute with the factor itself
    in the oracle's order (msolve of oracle/ipm_ref.py: forward, one correction, backward, one correction; one lane per row, two rows per
    lane above 64) instead of multiplying by inverted diagonal blocks.  Same status and iteration count as the oracle and as the default,
    objective to 1e-6 (y too where it is unique)."""
    b, A, ys, Xs, Zs = instances.planted_dense(n, min(m, n * (n + 1) // 2 - 1))
    core = ipm_ref.CoreProblem(b, [A])
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    monkeypatch.setenv("HIPSDP_SOLVE1", "0")
    dflt = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.setenv("HIPSDP_SMALL_SOLVE", "subst")
    sub = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    assert ref.status == 0 and dflt["info"].status == 0 and sub["info"].status == 0
    assert sub["info"].iterations == ref.iterations == dflt["info"].iterations
    # (objectives: with m close to the dimension of the matrix space the optimal y is not unique to this accuracy)
    assert abs(sub["info"].dobj - ref.dobj) <= 1e-6 * (1 + abs(ref.dobj)) and abs(sub["info"].dobj - dflt["info"].dobj) <= 1e-6 * (1 + abs(ref.dobj))
    if m <= n * (n + 1) // 4:
        assert np.max(np.abs(sub["y"] - ref.y)) <= 1e-6 and np.max(np.abs(sub["y"] - dflt["y"])) <= 1e-6
    assert not np.array_equal(sub["y"], dflt["y"])                    # (another algorithm: other last bits)


@pytest.mark.parametrize("sizes", [(100,), (70, 20)])
def test_failed_cholesky_check_takes_the_optimistic_step_back(gpu, sizes, monkeypatch):
    """Round 6: on the general path the rest of the step and the residual pass of the next iterate are queued behind the Cholesky check
    of the new X and Z, one read-back for both (csrc/ipm.hip, "optimistic").  A step beyond the boundary of the cone - forced here in
    iteration 2 (HIPSDP_TEST_OVERSTEP) - fails the check: the step is halved, the optimistic part taken back, the dual residual
    recomputed.  Same optimum, same iteration count and the same number of failed checks as the form that reads the flags first
    (HIPSDP_NO_OPTIMISTIC=1); blocks of at most 64 rows keep their trial iterate in a second buffer that is swapped in and back."""
    rng = np.random.default_rng(23)
    m = 40
    blocks = []
    for n in sizes:
        A = rng.standard_normal((m + 1, n, n))
        A = A + A.transpose(0, 2, 1)
        A[0] = -np.eye(n) * 3.0 + 0.1 * A[0]
        blocks.append(A)
    D = np.concatenate([np.eye(m), -np.eye(m)])
    c = np.concatenate([-2.0 * np.ones(m), -2.0 * np.ones(m)])
    core = ipm_ref.CoreProblem(rng.standard_normal(m), blocks, D, c)
    monkeypatch.setenv("HIPSDP_SOLVE1", "0")
    plain = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    assert plain["info"].status == 0 and plain["info"].chol_fail == 0
    monkeypatch.setenv("HIPSDP_TEST_OVERSTEP", "1.6")
    opt = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.setenv("HIPSDP_NO_OPTIMISTIC", "1")
    old = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    assert opt["info"].status == 0 and old["info"].status == 0
    assert opt["info"].chol_fail >= 1 and opt["info"].chol_fail == old["info"].chol_fail
    assert opt["info"].iterations == old["info"].iterations
    assert abs(opt["info"].dobj - plain["info"].dobj) <= 1e-6 * (1 + abs(plain["info"].dobj))
    assert np.max(np.abs(opt["y"] - old["y"])) <= 1e-7
    ok, det = checker.certificate(core, opt["y"], opt["X"], opt["lp"][0], TOL, TOL)
    assert ok, det


@pytest.mark.parametrize("mode", ["K3", "R", "U"])
def test_sharded_forms_with_several_blocks_and_lp_rows(gpu, mode, monkeypatch):
    """blocks of different sizes (70 crosses the 64 boundary, 20 leaves slices of the column form empty) plus LP rows: the
    sharded Schur forms add the blocks' contributions, then the LP term, exactly like the default assembly"""
    rng = np.random.default_rng(17)
    m = 30
    blocks = []
    for n in (70, 20):
        A = rng.standard_normal((m + 1, n, n))
        A = A + A.transpose(0, 2, 1)
        A[0] = -np.eye(n) * 3.0 + 0.1 * A[0]
        blocks.append(A)
    D = np.concatenate([rng.standard_normal((4, m)), np.eye(m), -np.eye(m)])
    c = np.concatenate([-np.ones(4) * 4.0, -2.0 * np.ones(m), -2.0 * np.ones(m)])
    core = ipm_ref.CoreProblem(rng.standard_normal(m), blocks, D, c)
    base = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.setenv("HIPSDP_SCHUR", mode)
    alt = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    monkeypatch.delenv("HIPSDP_SCHUR")
    assert base["info"].status == 0 and alt["info"].status == 0 and alt["info"].iterations == base["info"].iterations
    assert np.max(np.abs(alt["y"] - base["y"])) <= 1e-8
    ok, det = checker.certificate(core, alt["y"], alt["X"], alt["lp"][0], TOL, TOL)
    assert ok, det


def test_full_size_t1_properties(gpu):
    """BASELINE's 1-GPU target size (n = 1000, m = 2000: 16 GB of constraint matrices generated in HBM): size-independent
    properties only - planted optimum value on both sides, the y-side acceptance test (lambda_min of sum A_i y_i - A_0 on
    the device, what sdpsolchecker.c:201-257 checks), X psd, and run-to-run bitwise reproducibility."""
    n, m = 1000, 2000
    Q, _ = np.linalg.qr(instances.counter_normal(20240 + 1000003, np.arange(n * n, dtype=np.uint64)).reshape(n, n))
    r = n // 4
    ev = 1.0 + instances.counter_uniform(20240 + 2000003, np.arange(n, dtype=np.uint64))
    Xs = (Q * np.where(np.arange(n) < r, ev, 0.0)) @ Q.T
    Zs = (Q * np.where(np.arange(n) < r, 0.0, ev)) @ Q.T
    ys = 2.0 * instances.counter_uniform(20240 + 3000003, np.arange(m, dtype=np.uint64)) - 1.0
    s = gpu.Solver(0)
    s.set_shape(m, [n], 0)
    b = s.gen_planted(n, m, 20240, Xs, Zs, ys)
    info = s.solve(gaptol=1e-5, feastol=1e-5)
    y1, X1 = s.y(), s.X(0)
    opt = float(b @ ys)
    assert info.status == 0
    assert abs(info.dobj - opt) <= TOL * (1 + abs(opt)) and abs(info.pobj - opt) <= TOL * (1 + abs(opt))
    assert abs(float(b @ y1) - opt) <= TOL * (1 + abs(opt))
    lmin, viol = s.check_y(y1)
    assert lmin[0] >= -TOL and viol == 0.0
    assert np.linalg.eigvalsh(X1)[0] >= -TOL
    info2 = s.solve(gaptol=1e-5, feastol=1e-5)
    assert np.array_equal(s.y(), y1) and info2.iterations == info.iterations
    assert info.schur_flops / max(info.schur_seconds, 1e-12) > 30e12           # north_star: >= 30 % of the FP64 matrix peak
    s.close()


def test_full_size_c4_properties(gpu):
    """BASELINE configs[3] (n = 2000, m = 4000) on ONE device: 128 GB of constraint matrices generated in HBM + their 64 GB packed
    copy; the W formulation runs in column slices because T and W (2 x 128 GB) do not fit beside them.  About 20 s per solve, so
    one solve only; size-independent properties: planted optimum on both sides, the y-side acceptance test with its rigorous
    certificate (hipsdp_check_y: Cholesky of Z(y) - sigma I, exact eigenvalue when that fails), X psd, primal residual."""
    n, m = 2000, 4000
    Q, _ = np.linalg.qr(instances.counter_normal(20240 + 1000003, np.arange(n * n, dtype=np.uint64)).reshape(n, n))
    r = n // 4
    ev = 1.0 + instances.counter_uniform(20240 + 2000003, np.arange(n, dtype=np.uint64))
    Xs = (Q * np.where(np.arange(n) < r, ev, 0.0)) @ Q.T
    Zs = (Q * np.where(np.arange(n) < r, 0.0, ev)) @ Q.T
    ys = 2.0 * instances.counter_uniform(20240 + 3000003, np.arange(m, dtype=np.uint64)) - 1.0
    s = gpu.Solver(0)
    try:
        s.set_shape(m, [n], 0)
        b = s.gen_planted(n, m, 20240, Xs, Zs, ys)
        info = s.solve(gaptol=1e-5, feastol=1e-5)
        y1, X1 = s.y(), s.X(0)
        opt = float(b @ ys)
        assert info.status == 0
        assert abs(info.dobj - opt) <= TOL * (1 + abs(opt)) and abs(info.pobj - opt) <= TOL * (1 + abs(opt))
        assert abs(float(b @ y1) - opt) <= TOL * (1 + abs(opt))
        assert info.pinf <= 1e-5 and info.dabs <= 1e-5 and info.gap <= 1e-5
        lmin, viol = s.check_y(y1)
        assert lmin[0] >= -TOL and viol == 0.0
        assert np.linalg.eigvalsh(X1)[0] >= -TOL
        assert abs(np.sum(X1 * Zs)) <= 1e-2 * (1 + abs(opt))          # X is (nearly) complementary to the planted Z*
        assert info.schur_flops / max(info.schur_seconds, 1e-12) > 30e12
    finally:
        s.close()


def test_w_formulation_in_slices_when_the_workspace_is_small(gpu):
    """when T and W do not fit the workspace budget as a whole (n = 2000, m = 4000 on one GPU: 2 x 128 GB) the engine keeps
    the cheaper W formulation by working through column slices one after the other; forced here with a tiny budget"""
    b, A, ys, Xs, Zs = instances.planted_dense(100, 60)
    core = ipm_ref.CoreProblem(b, [A])
    base = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    for gb in (0.004, 0.0012):                                   # 4 MB -> 3 slices, 1.2 MB -> the 16-column granularity
        alt = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6, ws_gbytes=gb)
        assert alt["info"].status == 0 and alt["info"].iterations == base["info"].iterations
        assert np.max(np.abs(alt["y"] - base["y"])) <= 1e-8


def test_identity_start_assembly_with_a_chunked_workspace_smaller_than_two_slabs(gpu):
    """with a chunked workspace (budget below the whole T, W pair) the split-K slabs hold sk m1 cols doubles, and
    for m1 > 8 cols that is less than the two m1 x m1 slabs the XCD-sliced Gram product of the cold start's first assembly
    (M = A A^T) asked for: the product must then go straight into M.  m1 = 1101, cols = 128: 16 * 1101 * 128 < 2 * 1101^2."""
    b, A, ys, Xs, Zs = instances.planted_dense(128, 1100)
    core = ipm_ref.CoreProblem(b, [A])
    base = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6)
    alt = gpu_solve(gpu, core, gaptol=1e-6, feastol=1e-6, ws_gbytes=0.01)
    assert base["info"].status == 0 and alt["info"].status == 0
    assert alt["info"].iterations == base["info"].iterations
    assert np.max(np.abs(alt["y"] - base["y"])) <= 1e-7
    opt = float(b @ ys)
    assert abs(alt["info"].dobj - opt) <= 1e-5 * (1 + abs(opt))
