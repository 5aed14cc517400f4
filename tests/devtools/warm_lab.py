#!/usr/bin/env python3
"""warm_lab.py - developer tool: iterations of a warm start (parent optimum blended with a scaled identity, the relax_sdp.c
recipe) against the cold start, for a perturbed copy of the problem (a branching-like change of the objective / bounds)"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness")); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, ipm_ref, instances, sdpa_io
def solve(core, start=None):
    s = hb.Solver(0); s.load_core(core)
    if start is not None: s.set_start(*start)
    info = s.solve(gaptol=1e-6, feastol=1e-6)
    out = (info, s.y(), [s.X(k) for k in range(len(core.blocks))], [s.Z(k) for k in range(len(core.blocks))], s.lp()); s.close(); return out
inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", "example_TT.dat-s.gz"))
D, c = sdpa_io.lp_dense(inst)
core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
info, y, X, Z, (x, z) = solve(core)
print("parent: status %d iterations %d" % (info.status, info.iterations))
# child: tighten one LP row's constant (like a bound change)
rng = np.random.default_rng(3)
obj2 = inst.obj * (1 + 0.05 * rng.standard_normal(len(inst.obj)))
c2 = c.copy(); c2[5] += 0.02
child = ipm_ref.CoreProblem(obj2, sdpa_io.dense_blocks(inst), D, c2)
ic = solve(child)[0]
print("child cold: status %d iterations %d" % (ic.status, ic.iterations))
for f in (0.5, 0.2, 0.05, 0.01):
    sc = max(1.0, max(np.abs(Xk).max() for Xk in X))
    st = (y * (1 - f), [(1 - f) * Xk + f * sc * np.eye(Xk.shape[0]) for Xk in X], [(1 - f) * Zk + f * sc * np.eye(Zk.shape[0]) for Zk in Z],
          (1 - f) * x + f * sc, (1 - f) * z + f * sc)
    iw = solve(child, st)[0]
    print("child warm f=%.2f: warm_started %d status %d iterations %d" % (f, iw.warm_started, iw.status, iw.iterations))
