#!/usr/bin/env python3
"""mkp_dbg.py [instance] - developer tool: solve a golden instance on the GPU engine and print the full certificate check"""
import os, sys, json, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, ipm_ref, checker, sdpa_io
import test_gpu_ipm as T
name = sys.argv[1] if len(sys.argv) > 1 else "example_MkP.dat-s.gz"
inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", name))
D, c = sdpa_io.lp_dense(inst)
core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
g = T.gpu_solve(hb, core, gaptol=1e-6, feastol=1e-6)
print("ref status", ref.status, "iters", ref.iterations, "dobj", ref.dobj)
print("gpu status", g["info"].status, "iters", g["info"].iterations, "dobj", g["info"].dobj, "pinf", g["info"].pinf, "dabs", g["info"].dabs, "gap", g["info"].gap)
ok, det = checker.certificate(core, g["y"], g["X"], g["lp"][0], 1e-5, 1e-5)
print(ok); print(json.dumps(det, indent=1, default=str))
ok, det = checker.certificate(core, ref.y, ref.X, ref.x, 1e-5, 1e-5)
print("oracle:", ok); print(json.dumps(det, indent=1, default=str))
