"""CPU tests of the boundary: the shared library loads and exports every symbol the headers declare; the parts of the
solver interface that need no device (lifecycle, parameters, penalty helpers, time-limit pre-check, error conventions)
behave like the reference backends.  No compute call is made here."""
import ctypes as C
import os
import re
import time
import pytest

from conftest import ROOT
import sdpi_call
import sdpi_prepare

HDR_DIR = os.path.join(ROOT, "include")


def declared_symbols(header, prefix):
    txt = open(os.path.join(HDR_DIR, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_library_exports_every_declared_symbol(hb):
    lib = hb.lib()
    names = declared_symbols("sdpisolver_hip.h", "SCIPsdpiSolver") + declared_symbols("hipsdp.h", "hipsdp_")
    names += declared_symbols("lapack_interface_hip.h", "SCIPlapack") if os.path.exists(os.path.join(HDR_DIR, "lapack_interface_hip.h")) else []
    assert len([n for n in names if n.startswith("SCIPsdpiSolver")]) == 53      # sdpisolver.h:79-724
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the test / bench entry points (include/hipsdp_units.h) live in a library of their own and NOT in the product library
    unit_names = declared_symbols("hipsdp_units.h", "hipsdp_")
    ulib = hb.ulib()
    assert not [n for n in unit_names if not hasattr(ulib, n)]
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", hb.LIBPATH], stdout=subprocess.PIPE, text=True).stdout
    product = set(re.findall(r"\b(hipsdp_\w+)\b", exported))
    assert not (product & set(unit_names)), sorted(product & set(unit_names))
    assert product == set(declared_symbols("hipsdp.h", "hipsdp_")), sorted(product ^ set(declared_symbols("hipsdp.h", "hipsdp_")))


def test_name_and_static_answers(hb):
    lib = hb.lib()
    lib.SCIPsdpiSolverGetSolverName.restype = C.c_char_p
    assert lib.SCIPsdpiSolverGetSolverName() == b"HIPSDP"
    assert lib.SCIPsdpiSolverDoesWarmstartNeedPrimal() == 1
    assert lib.SCIPsdpiSolverGetDefaultSdpiSolverNpenaltyIncreases() > 0
    lib.SCIPsdpiSolverInfinity.restype = C.c_double
    assert lib.SCIPsdpiSolverInfinity(None) == 1e20                  # sdpisolver_dsdp.c:2634-2639
    assert lib.SCIPsdpiSolverIsInfinity(None, C.c_double(-1e20)) == 1
    assert lib.SCIPsdpiSolverIsInfinity(None, C.c_double(1e19)) == 0


def test_lifecycle_params_and_leak_check(hb):
    lib = hb.lib()
    lib.hipsdp_compat_mem_used.restype = C.c_longlong
    base = lib.hipsdp_compat_mem_used()
    s = sdpi_call.SdpiSolver(lib)
    # defaults at create: sdpisolver_dsdp.c:558-568
    assert s.get_real(0) == (sdpi_call.SCIP_OKAY, 1e-9)
    assert s.get_real(1) == (sdpi_call.SCIP_OKAY, 1e-6)
    assert s.get_real(3) == (sdpi_call.SCIP_OKAY, 1e-6)
    assert s.get_real(7) == (sdpi_call.SCIP_OKAY, 1e5)
    assert s.get_real(4) == (sdpi_call.SCIP_OKAY, 1e20)
    assert s.set_real(1, 1e-5) == sdpi_call.SCIP_OKAY and s.get_real(1)[1] == 1e-5
    assert s.set_real(10, 3.0) == sdpi_call.SCIP_OKAY                 # lambdastar accepted and ignored
    assert s.set_real(8, 1.0) == sdpi_call.SCIP_PARAMETERUNKNOWN      # unknown ids: sdpi.c:167-195 tolerates exactly this
    assert s.set_int(5, 1) == sdpi_call.SCIP_OKAY
    assert s.set_int(14, 1) == sdpi_call.SCIP_PARAMETERUNKNOWN
    # penalty helpers: clamp(1e4 * maxcoeff, 1e5, 1e12) and min(1e6 * Gamma, 1e15) (sdpisolver_dsdp.c:2803-2869)
    v = C.c_double(0)
    lib.SCIPsdpiSolverComputePenaltyparam(s.h, C.c_double(1.0), C.byref(v)); assert v.value == 1e5
    lib.SCIPsdpiSolverComputePenaltyparam(s.h, C.c_double(1e3), C.byref(v)); assert v.value == 1e7
    lib.SCIPsdpiSolverComputePenaltyparam(s.h, C.c_double(1e10), C.byref(v)); assert v.value == 1e12
    lib.SCIPsdpiSolverComputeMaxPenaltyparam(s.h, C.c_double(1e5), C.byref(v)); assert v.value == 1e11
    lib.SCIPsdpiSolverComputeMaxPenaltyparam(s.h, C.c_double(1e12), C.byref(v)); assert v.value == 1e15
    # nothing solved yet: predicates FALSE, getters SCIP_LPERROR (CHECK_IF_SOLVED, sdpisolver_dsdp.c:145-164)
    assert not s.flag("WasSolved") and not s.flag("IsAcceptable") and not s.flag("IsOptimal")
    assert s.internal_status() == -1
    o = C.c_double(0)
    assert lib.SCIPsdpiSolverGetObjval(s.h, C.byref(o)) == sdpi_call.SCIP_LPERROR
    assert s.settings_used() == -1
    ok = C.c_uint(1)
    lib.SCIPsdpiSolverGetPreoptimalSol(s.h, C.byref(ok), None, -1, None, None, None, None)
    assert ok.value == 0
    assert lib.SCIPsdpiSolverReadSDP(s.h, b"x") == sdpi_call.SCIP_LPERROR
    lib.SCIPsdpiSolverIncreaseCounter(s.h)
    lib.SCIPsdpiSolverResetCounter(s.h)
    s.free()
    assert lib.hipsdp_compat_mem_used() == base                       # checksdpi.c:117


def test_time_limit_precheck_needs_no_device(hb):
    """remaining time <= 0: timelimit flags set, solved = FALSE, SCIP_OKAY, zero iterations (sdpisolver_dsdp.c:879-892)"""
    lib = hb.lib()
    lib.SDPIclockGetTime.restype = C.c_double
    clk = C.c_void_p()
    assert lib.SDPIclockCreate(C.byref(clk)) == sdpi_call.SCIP_OKAY
    lib.SDPIclockStart(clk)
    time.sleep(0.02)
    s = sdpi_call.SdpiSolver(lib)
    prob = sdpi_prepare.SdpiProblem([-3, -1], [0, 0], [1e20, 1e20], [], [(-1e20, 10, {0: 2, 1: 1}), (-1e20, 15, {0: 1, 1: 3})])
    rc, _, _ = s.solve(sdpi_prepare.prepare(prob), timelimit=0.01, clock=clk)
    assert rc == sdpi_call.SCIP_OKAY
    assert s.flag("IsTimelimExc") and not s.flag("WasSolved") and not s.flag("IsAcceptable")
    assert s.internal_status() == 5
    assert s.iterations() == 0 and s.sdpcalls() == 0
    s.free()
    lib.SDPIclockStop(clk)
    assert lib.SDPIclockGetTime(clk) >= 0.02
    lib.SDPIclockFree(C.byref(clk))


def test_engine_refuses_to_run_without_device(hb):
    """no CPU fallback: on a machine without a GPU the engine reports HIPSDP_ERR_NODEVICE instead of computing"""
    if hb.device_count() > 0:
        pytest.skip("a device is present; covered by the gpu tests")
    with pytest.raises(RuntimeError):
        hb.Solver(0)
    with pytest.raises(RuntimeError):
        import numpy as np
        hb.dgemm(np.eye(2), np.eye(2))


def test_marshalling_restatement_consistency():
    """sdpi_prepare.to_core on a problem with a fixed variable, an emptied row/column and a ranged row"""
    import numpy as np
    blocks = [dict(n=3, vars={0: [(0, 0, 1.0)], 1: [(1, 1, 1.0), (2, 2, 2.0)], 2: [(1, 0, 0.5)]}, const=[(0, 0, -1.0)])]
    lp = [(1.0, 4.0, {0: 1.0, 1: 1.0, 2: 1.0}), (-1e20, 3.0, {1: 2.0})]
    prob = sdpi_prepare.SdpiProblem([1, 2, 3], [0, 1, -1e20], [5, 1, 1e20], blocks, lp)      # variable 1 fixed to 1
    P = sdpi_prepare.prepare(prob)
    assert list(P.indchanges[0]) == [0, 0, 0]             # the fixed variable's diagonal moved into the constant part
    assert sorted(P.sdpconst[0]) == [(0, 0, -1.0), (1, 1, -1.0), (2, 2, -2.0)]
    assert list(P.lpindchanges) == [0, -1]                # second row has no active nonzero left
    assert P.lplhs[0] == 0.0 and P.lprhs[0] == 3.0        # shifted by the fixed variable
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    assert maps["active"] == [0, 2] and list(b) == [1.0, 3.0]
    assert D.shape == (2 + 2, 2)                          # two LP sides + lb, ub of variable 0
    assert np.allclose(blk[0][0], np.diag([-1.0, -1.0, -2.0]))


def test_coefficient_tightening_textbook_row():
    """sdpi.c:812-1129 restated: 3 x + 2 y <= 4 over binaries becomes x + y <= 1 (Achterberg, Algorithm 10.1)"""
    act, lhs, rhs, lred, rred, nchg = sdpi_prepare.tighten_row_coefs([0.0, 0.0], [1.0, 1.0], [(0, 3.0), (1, 2.0)], -1e20, 4.0, [True, True])
    assert dict(act) == {0: 1.0, 1: 1.0} and rhs == 1.0 and lhs == -1e20 and nchg == 2 and lred and not rred
    # continuous variables and equations are left alone; a row redundant on both sides is reported as such
    assert sdpi_prepare.tighten_row_coefs([0.0, 0.0], [1.0, 1.0], [(0, 3.0), (1, 2.0)], -1e20, 4.0, [False, False])[5] == 0
    assert sdpi_prepare.tighten_row_coefs([0.0, 0.0], [1.0, 1.0], [(0, 3.0), (1, 2.0)], 4.0, 4.0, [True, True])[5] == 0
    out = sdpi_prepare.tighten_row_coefs([0.0, 0.0], [1.0, 1.0], [(0, 3.0), (1, 2.0)], -1.0, 6.0, [True, True])
    assert out[3] and out[4]


def test_coefficient_tightening_keeps_the_points_with_integral_values():
    """brute force over small boxes: with the integral variables at integer values (continuous ones sampled on a grid) a point
    satisfies the tightened row exactly when it satisfied the original one"""
    import itertools
    import numpy as np
    rng = np.random.default_rng(42)
    changed = 0
    for trial in range(300):
        k = int(rng.integers(2, 5))
        lb = rng.integers(-2, 1, k).astype(float)
        ub = lb + rng.integers(1, 4, k)
        isint = [bool(b) for b in rng.integers(0, 2, k)]
        coef = rng.integers(-5, 6, k).astype(float)
        coef[coef == 0.0] = 1.0
        act = [(j, coef[j]) for j in range(k)]
        lo_act = sum(c * (lb[j] if c > 0 else ub[j]) for j, c in act)
        hi_act = sum(c * (ub[j] if c > 0 else lb[j]) for j, c in act)
        lhs = -1e20 if rng.random() < 0.5 else float(np.floor(lo_act + rng.random() * (hi_act - lo_act)))
        rhs = 1e20 if (lhs > -1e20 and rng.random() < 0.5) else float(np.ceil(max(lhs if lhs > -1e20 else lo_act, lo_act) + rng.random() * (hi_act - lo_act)))
        new, nlhs, nrhs, lred, rred, nchg = sdpi_prepare.tighten_row_coefs(lb, ub, act, lhs, rhs, isint)
        changed += nchg
        nd = dict(new)
        grids = [np.arange(lb[j], ub[j] + 0.5) if isint[j] else np.linspace(lb[j], ub[j], 5) for j in range(k)]
        for pt in itertools.product(*grids):
            old = lhs - 1e-9 <= sum(c * pt[j] for j, c in act) <= rhs + 1e-9
            val = sum(nd.get(j, 0.0) * pt[j] for j in range(k))
            newok = (nlhs <= -1e20 or nlhs - 1e-9 <= val) and (nrhs >= 1e20 or val <= nrhs + 1e-9)
            if lred and rred:
                assert old                                   # a redundant row holds on the whole box
            else:
                assert old == newok, (trial, pt, act, lhs, rhs, new, nlhs, nrhs)
    assert changed > 20          # the random rows do exercise the tightening


def test_prepare_tightens_rows_with_integral_variables():
    prob = sdpi_prepare.SdpiProblem([1, 1, 1], [0, 0, 0], [1, 1, 5], [], [(-1e20, 4.0, {0: 3.0, 1: 2.0}), (-1e20, 9.0, {0: 1.0, 2: 1.0})],
                                    isintegral=[True, True, False])
    P = sdpi_prepare.prepare(prob)
    assert P.status == 'ok' and P.nchgcoefs == 2
    assert list(P.lpindchanges) == [0, -1]                    # row 1 (x0 + y2 <= 9 with x0 <= 1, y2 <= 5) is redundant and removed
    assert P.lprhs[0] == 1.0 and list(P.lpval[:2]) == [1.0, 1.0] and list(P.lpind[:2]) == [0, 1]


def test_communicator_from_the_environment_without_a_launch(hb):
    """hipsdp_comm_from_env (what sdpisolver_hip.c calls when it creates its engine): no WORLD_SIZE / HIPSDP_WORLD, or a world of
    one, means a single GPU - no communicator, no device call; a world without a rendezvous (neither HIPSDP_COMM_FILE nor
    HIPSDP_COMM_SHM) is an error, not a hang.  Run in child processes: the communicator is process-wide state."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import ctypes as C, importlib.util, os, sys
        spec = importlib.util.spec_from_file_location("hipsdp_binding", sys.argv[1])
        hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
        comm, rank, world = C.c_void_p(), C.c_int(-1), C.c_int(-1)
        rc = hb.lib().hipsdp_comm_from_env(0, C.byref(comm), C.byref(rank), C.byref(world))
        print(rc, comm.value, rank.value, world.value)
    """)
    binding = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scip-sdp_amd", "binding.py")
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "HIPSDP_WORLD", "HIPSDP_RANK", "HIPSDP_COMM_FILE", "HIPSDP_COMM_SHM")}
    out = subprocess.run([sys.executable, "-c", code, binding], env=base, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split() == ["0", "None", "0", "1"], (out.stdout, out.stderr)
    out = subprocess.run([sys.executable, "-c", code, binding], env=dict(base, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert out.stdout.split() == ["0", "None", "0", "1"]
    out = subprocess.run([sys.executable, "-c", code, binding], env=dict(base, HIPSDP_WORLD="4", HIPSDP_RANK="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split()[0] != "0" and out.stdout.split()[1] == "None", (out.stdout, out.stderr)
