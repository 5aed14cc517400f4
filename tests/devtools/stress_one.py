#!/usr/bin/env python3
"""stress_one.py seed - developer tool: one problem of stress_gpu.py, verbose, optionally with another library (HIPSDP_LIB)"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
if os.environ.get("HIPSDP_LIB"):
    hb.LIBPATH = os.environ["HIPSDP_LIB"]
import numpy as np, ipm_ref
import stress_gpu
stress_gpu.hb = hb
seed = int(sys.argv[1])
core, kind = stress_gpu.rand_core(np.random.default_rng(seed))
ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, verbose=len(sys.argv) > 2))
s = hb.Solver(0); s.load_core(core)
info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, verbose=1 if len(sys.argv) > 2 else 0)
print("oracle status %d it %d dobj %.9g | gpu status %d it %d dobj %.9g pinf %.2e dabs %.2e gap %.2e" % (ref.status, ref.iterations, ref.dobj, info.status, info.iterations, info.dobj, info.pinf, info.dabs, info.gap))
