"""developer script: the one-launch solve of an instance's root node repeated: device time of the kernel (its own wall clock), engine
time (hipsdp_solve: argument block, launch, wait for the result block) and the time of the whole load + solve sequence of the binding"""
import sys, os, time, importlib.util
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
s.load_core(core)
for r in range(12):
    t0 = time.perf_counter()
    info = s.solve(gaptol=1e-6, feastol=1e-6)
    t1 = time.perf_counter()
    out, _ = s.solve1_trace(1)
    print("solve %2d: %d iterations, kernel %.1f us, hipsdp_solve %.1f us (engine clock %.1f us)" % (r, info.iterations, out[43] / 100.0, (t1 - t0) * 1e6, info.solve_seconds * 1e6))
