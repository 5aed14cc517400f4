"""developer script: device cycles per phase of the one-launch kernel (HIPSDP_SOLVE1_PROF=1, printed by the engine on stderr) on the
synthetic shapes of solve1_sizes.py"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests'), os.path.dirname(os.path.abspath(__file__))]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
from solve1_sizes_lib import core_of
for sizes, m, q in [([16], 40, 40), ([24], 40, 40), ([32], 48, 40), ([12, 12, 12], 40, 40), ([30, 30], 50, 20)]:
    core = core_of(sizes, m, q, 5)
    s = hb.Solver(0); s.load_core(core)
    s.solve(gaptol=1e-6, feastol=1e-6)
    os.environ["HIPSDP_SOLVE1_PROF"] = "1"
    print("== blocks %s m %d q %d" % (sizes, m, q), file=sys.stderr, flush=True)
    s.solve(gaptol=1e-6, feastol=1e-6)
    os.environ["HIPSDP_SOLVE1_PROF"] = "0"
    s.close()
