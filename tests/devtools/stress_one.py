#!/usr/bin/env python3
"""stress_one.py SEED [settings] - developer tool: one problem of the stress family (tests/stress_cases.py) on engine and oracle, verbose"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness")); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, ipm_ref, stress_cases
seed = int(sys.argv[1]); lv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
core, kind = stress_cases.rand_core(np.random.default_rng(seed))
print("seed", seed, "ns", [A.shape[1] for A in core.blocks], "m", core.m, "q", core.q, "kind", kind, flush=True)
ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, settings=lv, verbose=True))
print("oracle status", ref.status, "iterations", ref.iterations, flush=True)
s = hb.Solver(0); s.load_core(core)
info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, settings=lv, verbose=1)
print("engine status", info.status, "iterations", info.iterations, "chol_fail", info.chol_fail)
s.close()
