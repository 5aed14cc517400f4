"""developer script: the root node of one reference instance through the one-launch kernel a few times (HIPSDP_SOLVE1_PROF=1 prints
the cycles per phase)"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref, sdpa_io
name = sys.argv[1] if len(sys.argv) > 1 else "example_TT.dat-s.gz"
inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", name)); D, c = sdpa_io.lp_dense(inst)
core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
s = hb.Solver(0)
for r in range(3):
    s.load_core(core); info = s.solve(gaptol=1e-6, feastol=1e-6)
print(name, "status", info.status, "iterations", info.iterations, "solve %.1f us" % (info.solve_seconds * 1e6), "path", s.solve_path())
