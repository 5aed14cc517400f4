#!/usr/bin/env python3
"""spmd_worker.py INSTANCE OUT.json - one copy of an SPMD host run (test helper).

The same branch-and-bound over SCIPsdpiSolverLoadAndSolve that tests/test_gpu_bnb.py runs, started as N identical processes with
HIPSDP_WORLD / HIPSDP_RANK / HIPSDP_COMM_SHM in the environment: the backend (sdpisolver_hip.c) joins the process-wide
communicator when it creates its engine and every node SDP is sharded over the ranks.  No call in this file knows about ranks -
this is what N copies of SCIP-SDP linked against libhipsdp.so do."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hb)
import bnb
import sdpa_io
import sdpi_call


def main():
    name, out = sys.argv[1], sys.argv[2]
    inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", name))
    prob = bnb.instance_to_sdpi(inst)
    s = sdpi_call.SdpiSolver(hb.lib())
    for par in (1, 2, 3):
        assert s.set_real(par, 1e-6) == sdpi_call.SCIP_OKAY
    stats = dict(calls=0, iters=0)

    def solve(P):
        rc, _, _ = s.solve(P)
        assert rc == sdpi_call.SCIP_OKAY
        stats["calls"] += 1
        stats["iters"] += s.iterations()
        if s.flag("IsDualInfeasible"):
            return bnb.NodeResult('infeasible')
        if s.flag("IsDualUnbounded"):
            return bnb.NodeResult('unbounded')
        if not s.flag("IsOptimal"):
            return bnb.NodeResult('failed')
        rc, obj, y = s.dual_sol()
        return bnb.NodeResult('optimal', obj, y)

    best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, solve)
    s.free()
    json.dump(dict(best=best, y=None if y is None else [float(v) for v in y], nodes=nodes, failed=failed, calls=stats["calls"],
                   iters=stats["iters"]), open(out, "w"))


if __name__ == "__main__":
    main()
