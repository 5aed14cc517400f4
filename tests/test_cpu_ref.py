"""CPU: the plain-C restatement of the iteration for B&B-sized problems (oracle/cpu_ref.c - the compiled CPU baseline of bench.py's
bnb leg) against the numpy oracle on the reference-held cases and instances: same status, same iteration count, same y.  Both are test
infrastructure; the pinned side is oracle/ipm_ref.py (tests/test_oracle_golden.py)."""
import json
import os
import numpy as np
import pytest

import cpu_ref
import ipm_ref
import sdpa_io
import sdpi_prepare
from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))["cases"]


def case_core(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    P = sdpi_prepare.prepare(sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp))
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    return ipm_ref.CoreProblem(b, blk, D, c)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_c_restatement_matches_the_numpy_oracle_on_reference_cases(case):
    core = case_core(case)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
    info, y = cpu_ref.solve(core, gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
    assert info.status == ref.status and info.iterations == ref.iterations
    ytol = 1e-3 if ref.status == ipm_ref.STATUS_PDINF else 1e-7
    assert np.allclose(y, ref.y, atol=ytol, rtol=ytol)


@pytest.mark.parametrize("name", ["example_small.dat-s", "example_TT.dat-s.gz", "example_inf.dat-s", "example_tightenmatrices.dat-s",
                                  "example_MkP.dat-s.gz"])
def test_c_restatement_matches_the_numpy_oracle_on_reference_instances(name):
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", name))
    D, c = sdpa_io.lp_dense(inst)
    core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
    info, y = cpu_ref.solve(core, gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
    assert info.status == ref.status == 0 and info.iterations == ref.iterations
    assert abs(info.dobj - ref.dobj) <= 1e-9 * (1 + abs(ref.dobj))
    assert np.max(np.abs(y - ref.y)) <= 1e-7 * (1 + np.max(np.abs(ref.y)))


@pytest.mark.parametrize("n,m", [(12, 20), (40, 60), (64, 150)])
def test_dense_c_restatement_on_the_host_blas_matches_the_numpy_oracle(n, m):
    """oracle/cpu_ref_dense.c (one dense block, level-3 BLAS from scipy's OpenBLAS, OpenMP: the compiled CPU figure beside bench.py's
    headline) against oracle/ipm_ref.py on planted problems: same status, same iteration count, objective and y to 1e-8; one thread and
    four give the same iteration count (the pieces of a product are summed in the same order whatever the thread count)."""
    import cpu_ref_dense
    import instances
    if cpu_ref_dense.openblas() is None:
        pytest.skip("no OpenBLAS with scipy_-prefixed symbols in this image")
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    ref = ipm_ref.hsd_solve(ipm_ref.CoreProblem(b, [A]), ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    for threads in (1, 4):
        info, y = cpu_ref_dense.solve(b, A, gaptol=1e-6, feastol=1e-6, threads=threads)
        assert info.status == ref.status == 0 and info.iterations == ref.iterations
        assert abs(info.dobj - ref.dobj) <= 1e-8 * (1 + abs(ref.dobj))
        assert np.max(np.abs(y - ref.y)) <= 1e-8
        assert abs(info.dobj - b @ ys) <= 1e-5 * (1 + abs(b @ ys))


def test_dense_c_restatement_certificate_status():
    """a problem without an optimum - the constraint matrices leave a direction y along which b^T y falls for ever while sum A_i y_i - A_0
    stays psd (status 2, the y-ray certificate of oracle/ipm_ref.py) -: the C restatement ends with the oracle's status after the same
    number of iterations (the certificate tests, the tau -> 0 zone and the normalisation of a ray are code the planted problems never
    reach)"""
    import cpu_ref_dense
    if cpu_ref_dense.openblas() is None:
        pytest.skip("no OpenBLAS with scipy_-prefixed symbols in this image")
    n, m = 6, 3
    rng = np.random.default_rng(5)
    A = np.zeros((m + 1, n, n))
    for i in range(1, m + 1):
        v = rng.standard_normal((n, 2))
        A[i] = -(v @ v.T)
    A[0] = np.eye(n)
    A[1] = -A[1]
    b = np.ones(m)
    ref = ipm_ref.hsd_solve(ipm_ref.CoreProblem(b, [A]), ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    info, y = cpu_ref_dense.solve(b, A, gaptol=1e-6, feastol=1e-6)
    assert ref.status == ipm_ref.STATUS_DUNB
    assert info.status == ref.status and info.iterations == ref.iterations
