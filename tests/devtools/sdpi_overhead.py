#!/usr/bin/env python3
"""sdpi_overhead.py [n m] - developer tool: wall time of SCIPsdpiSolverLoadAndSolve against the time inside the engine for a dense
mid-size node SDP (first call: upload of the master copy; second call: the device-resident copy is reused)."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import instances, sdpi_prepare, sdpi_call
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200
b, A, ys, Xs, Zs = instances.planted_dense(n, m)
ii, jj = np.tril_indices(n)
t0 = time.perf_counter()
blocks = [dict(n=n, vars={v: list(zip(ii.tolist(), jj.tolist(), A[v + 1][ii, jj].tolist())) for v in range(m)},
               const=list(zip(ii.tolist(), jj.tolist(), A[0][ii, jj].tolist())))]
P = sdpi_prepare.prepare(sdpi_prepare.SdpiProblem(b.tolist(), [-1e20] * m, [1e20] * m, blocks, []))
print("python-side preparation %.2f s (test harness, not the backend)" % (time.perf_counter() - t0))
s = sdpi_call.SdpiSolver(hb.lib())
for par in (1, 2, 3):
    s.set_real(par, 1e-5)
if os.environ.get("SDPINFO"):
    s.set_int(5, 1)
for call in range(3):
    t0 = time.perf_counter(); rc, _, _ = s.solve(P); t1 = time.perf_counter()
    print("call %d: rc %d  LoadAndSolve wall %.1f ms  engine %.1f ms  iterations %d  optimal %s" % (call, rc, 1e3 * (t1 - t0), 1e3 * s.opttime(),
          s.iterations(), s.flag("IsOptimal")))
# a different instance of the same shape on the same solver object: the master copy is uploaded again (no creation costs this time)
b2, A2, _, _, _ = instances.planted_dense(n, m, seed=777)
blocks2 = [dict(n=n, vars={v: list(zip(ii.tolist(), jj.tolist(), A2[v + 1][ii, jj].tolist())) for v in range(m)},
                const=list(zip(ii.tolist(), jj.tolist(), A2[0][ii, jj].tolist())))]
P2 = sdpi_prepare.prepare(sdpi_prepare.SdpiProblem(b2.tolist(), [-1e20] * m, [1e20] * m, blocks2, []))
for call in range(2):
    t0 = time.perf_counter(); rc, _, _ = s.solve(P2); t1 = time.perf_counter()
    print("new instance, call %d: rc %d  LoadAndSolve wall %.1f ms  engine %.1f ms" % (call, rc, 1e3 * (t1 - t0), 1e3 * s.opttime()))
s.free()
