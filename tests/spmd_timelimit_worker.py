#!/usr/bin/env python3
"""spmd_timelimit_worker.py OUT.json - one copy of an SPMD host run whose clocks disagree (test helper).
Every copy calls SCIPsdpiSolverLoadAndSolve with the SAME time limit, but the copy with HIPSDP_RANK = SLOW_RANK started its clock
SLOW_SECONDS earlier, so only that copy's clock is past the limit when the call begins.  The decision must be rank 0's on every
copy (sdpisolver_hip.c: timeLeft): nobody may leave the others alone in a collective."""
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hb)
import numpy as np
import sdpi_prepare
import sdpi_call


def main():
    out = sys.argv[1]
    rank = int(os.environ.get("HIPSDP_RANK", "0"))
    slow = int(os.environ.get("SLOW_RANK", "-1"))
    lib = hb.lib()
    lib.SDPIclockGetTime.restype = C.c_double
    rng = np.random.default_rng(5)
    n, nvars = 70, 90                                   # big enough to be sharded (HIPSDP_SHARD_MIN_FLOPS=0 in the test)
    vars_ = {}
    for v in range(nvars):
        G = rng.standard_normal((n, n))
        vars_[v] = [(r, c, float(G[r, c] + G[c, r])) for r in range(n) for c in range(r + 1) if (r + c + v) % 3 == 0]
    prob = sdpi_prepare.SdpiProblem(rng.standard_normal(nvars), [-1.0] * nvars, [1.0] * nvars,
                                    [dict(n=n, vars=vars_, const=[(i, i, -40.0) for i in range(n)])], [])
    P = sdpi_prepare.prepare(prob)
    s = sdpi_call.SdpiSolver(lib)
    for par in (1, 2, 3):
        s.set_real(par, 1e-6)
    # the engine and the communicator are created by the first call (the copies meet there): keep that out of the timed calls
    rc, _, _ = s.solve(P)
    assert rc == 1 and s.flag("IsOptimal")
    res = []
    for limit in (float(os.environ.get("TIME_LIMIT", "0.5")), 1e-3):
        clk = C.c_void_p()
        assert lib.SDPIclockCreate(C.byref(clk)) == 1
        lib.SDPIclockStart(clk)
        if rank == slow:
            time.sleep(float(os.environ.get("SLOW_SECONDS", "1.0")))
        else:
            time.sleep(0.05)
        rc, _, _ = s.solve(P, timelimit=limit, clock=clk)
        res.append(dict(rc=rc, solved=s.flag("WasSolved"), timelim=s.flag("IsTimelimExc"), optimal=s.flag("IsOptimal"),
                        obj=s.objval() if s.flag("IsOptimal") else None, clock=lib.SDPIclockGetTime(clk)))
        lib.SDPIclockFree(C.byref(clk))
    s.free()
    json.dump(res, open(out, "w"))


if __name__ == "__main__":
    main()
