"""GPU: a slice of the stress families of tests/devtools/stress_gpu.py in the driver-run suite (the full campaigns - hundreds of
problems per family - are builder runs, profiles/r0*_stress*.txt): random engine problems against the oracle - status, iteration
count, objective / y, certificates.  Family "mid": 1-2 dense or sparse blocks of 100-257 rows, 129-400 variables, up to 150 LP rows
(general kernels, MFMA tile paths, K-sliced Gram product); family "big": towards the bench size (blocks of 300-500 rows, 300-1000
variables: persistent GEMM paths); family "small": the sizes around the 64 / 128 boundaries of the single-launch kernels."""
import numpy as np
import pytest

import checker
import ipm_ref
from stress_cases import rand_core

pytestmark = pytest.mark.gpu


# engine and oracle within ONE iteration of each other (was two); the problems on which the last iterates sit at the
# tolerance and the two stop two iterations apart are listed by family and seed
ITERATIONS_OFF_BY_TWO = set()


def run(gpu, seed, monkeypatch, family):
    if family == "mid":
        monkeypatch.setenv("STRESS_BIG", "1")
    elif family == "big":
        monkeypatch.setenv("STRESS_BIG", "2")
    else:
        monkeypatch.delenv("STRESS_BIG", raising=False)
    core, kind = rand_core(np.random.default_rng(seed))
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
    s = gpu.Solver(0)
    s.load_core(core)
    info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
    y = s.y(); X = [s.X(k) for k in range(len(core.blocks))]; lp = s.lp()
    s.close()
    tag = "seed %d ns %s m %d q %d kind %d" % (seed, [A.shape[1] for A in core.blocks], core.m, core.q, kind)
    # a numerical failure on BOTH sides is an agreement (the caller's ladder / penalty formulation takes such nodes)
    if info.status >= 4 and ref.status >= 4:
        return
    assert info.status == ref.status, tag
    assert abs(info.iterations - ref.iterations) <= (2 if (family, seed) in ITERATIONS_OFF_BY_TWO else 1), tag
    if ref.status == ipm_ref.STATUS_OPTIMAL:
        assert abs(info.dobj - ref.dobj) <= 1e-5 * (1 + abs(ref.dobj)), tag
        # (y is unique only while the variables do not outnumber the dimensions of the matrix space)
        if kind in (0, 1) and 2 * core.m <= sum(A.shape[1] * (A.shape[1] + 1) // 2 for A in core.blocks):
            assert np.max(np.abs(y - ref.y)) <= 1e-6 * (1 + np.max(np.abs(ref.y))), tag
        ok, det = checker.certificate(core, y, X, lp[0], 1e-5, 1e-5)
        assert ok, (tag, det)
    elif ref.status in (ipm_ref.STATUS_DINF, ipm_ref.STATUS_PDINF):
        assert checker.farkas_dual_infeasible(core, X, lp[0], 1e-6)[0], tag


@pytest.mark.parametrize("seed", range(1000, 1040))
def test_mid_size_stress_slice(gpu, seed, monkeypatch):
    run(gpu, seed, monkeypatch, "mid")


@pytest.mark.parametrize("seed", range(2000, 2004))
def test_near_bench_size_stress_slice(gpu, seed, monkeypatch):
    run(gpu, seed, monkeypatch, "big")


@pytest.mark.parametrize("seed", range(0, 60))
def test_small_size_stress_slice(gpu, seed, monkeypatch):
    run(gpu, seed, monkeypatch, "small")
