#!/usr/bin/env python3
"""lp_heavy.py - developer tool: problems dominated by LP rows (no block / one small block, hundreds to thousands of rows) on the HIP
engine against the oracle."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, ipm_ref
bad = 0
for seed, (m, q, n) in enumerate([(50, 400, 0), (300, 1500, 0), (120, 3000, 0), (200, 1000, 12), (400, 2500, 40), (64, 5000, 70)]):
    rng = np.random.default_rng(seed)
    y0 = rng.uniform(-1, 1, m)
    D = rng.standard_normal((q, m)) * (rng.random((q, m)) < 0.2)
    D = np.vstack([D, np.eye(m), -np.eye(m)])                                  # a box keeps the problem bounded
    c = D @ y0 - rng.uniform(0.1, 2.0, D.shape[0])                             # y0 strictly feasible
    blocks = []
    if n:
        A = rng.standard_normal((m + 1, n, n)); A = (A + A.transpose(0, 2, 1)) / np.sqrt(2 * n)
        A[0] = np.tensordot(y0, A[1:], axes=(0, 0)) - np.eye(n)                # Z(y0) = I
        blocks = [A]
    core = ipm_ref.CoreProblem(rng.standard_normal(m), blocks, D, c)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
    s = hb.Solver(0); s.load_core(core); info = s.solve(gaptol=1e-6, feastol=1e-6); y = s.y(); s.close()
    ok = info.status == ref.status and abs(info.dobj - ref.dobj) <= 1e-6 * (1 + abs(ref.dobj)) and abs(info.iterations - ref.iterations) <= 1
    bad += 0 if ok else 1
    print("m %4d rows %5d block %3d: status %d/%d iterations %d/%d objective %.9g/%.9g max|y-yref| %.1e %s" % (m, D.shape[0], n, info.status,
          ref.status, info.iterations, ref.iterations, info.dobj, ref.dobj, np.max(np.abs(y - ref.y)), "" if ok else "  <-- DIFFERENT"), flush=True)
print("%d different" % bad)
