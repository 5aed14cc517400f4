#!/usr/bin/env python3
"""stress_gpu.py [count] [seed0] (STRESS_BIG=1: blocks of 100-257 rows, 129-400 variables) - developer tool: random engine problems (dense/sparse blocks across the 64 and 128 size
boundaries, LP rows, feasible / infeasible / unbounded mixes) on the HIP engine against the oracle: status, objective,
certificates."""
import os, sys, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness")); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, ipm_ref, checker

from stress_cases import rand_core

def main():
    nfail = [0, 0]
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t0 = time.time()
    for t in range(count):
        rng = np.random.default_rng(seed0 + t)
        core, kind = rand_core(rng)
        ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
        s = hb.Solver(0); s.load_core(core)
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
        if os.environ.get("STRESS_LADDER"):
            # what the backend does around the engine (sdpisolver_sdpa.cpp:1698-1795): a failed solve climbs to the medium, then
            # to the stable settings - on both sides
            for lv in (1, 2):
                if ref.status >= 4:
                    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, settings=lv))
                if info.status >= 4:
                    info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, settings=lv)
        y = s.y(); X = [s.X(k) for k in range(len(core.blocks))]; lp = s.lp(); s.close()
        msg = []
        nfail[0] += 1 if info.status >= 4 else 0
        nfail[1] += 1 if ref.status >= 4 else 0
        if info.status != ref.status:
            msg.append("status gpu %d oracle %d" % (info.status, ref.status))
        elif ref.status == ipm_ref.STATUS_OPTIMAL:
            if abs(info.dobj - ref.dobj) > 1e-5 * (1 + abs(ref.dobj)):
                msg.append("dobj gpu %.9g oracle %.9g" % (info.dobj, ref.dobj))
            ok, det = checker.certificate(core, y, X, lp[0], 1e-5, 1e-5)
            if not ok:
                msg.append("certificate fails: %s" % det)
        elif ref.status in (ipm_ref.STATUS_DINF, ipm_ref.STATUS_PDINF):
            if not checker.farkas_dual_infeasible(core, X, lp[0], 1e-6)[0]:
                msg.append("X-ray fails")
        if abs(info.iterations - ref.iterations) > 2 and not os.environ.get("STRESS_LADDER"):
            msg.append("iterations gpu %d oracle %d" % (info.iterations, ref.iterations))
        tag = "ns %s m %d q %d kind %d status %d it %d" % ([A.shape[1] for A in core.blocks], core.m, core.q, kind, info.status, info.iterations)
        if msg:
            bad += 1
            print("seed %d: %s :: %s" % (seed0 + t, tag, "; ".join(msg)), flush=True)
        elif os.environ.get("STRESS_LIST_FAIL") and (info.status >= 4 or ref.status >= 4):
            print("seed %d: %s :: both fail (gpu %d oracle %d)" % (seed0 + t, tag, info.status, ref.status), flush=True)
    print("%d problems, %d with differences, %.1f s; numerical failures: gpu %d oracle %d" % (count, bad, time.time() - t0, nfail[0], nfail[1]))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
