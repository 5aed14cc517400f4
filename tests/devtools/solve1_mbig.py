"""developer script: the one-launch solve of problems with 64 < m <= 128 (csrc/solve1_c64m.hip: the Schur matrix as a packed lower
triangle) against the general path: the root node of example_MkP and synthetic shapes up to m = 128; ms per interior-point iteration
on both paths, the phases of the kernel (device cycles) for the first"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests'), os.path.dirname(os.path.abspath(__file__))]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref, sdpa_io
from solve1_sizes_lib import core_of

inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", "example_MkP.dat-s.gz"))
D, c = sdpa_io.lp_dense(inst)
cases = [("example_MkP root", ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c))]
os.environ.setdefault("HIPSDP_SOLVE1_MAXM", "128")
for sizes, m, q in [([15], 70, 100), ([15], 90, 200), ([15], 105, 240), ([16], 120, 100), ([20], 128, 60), ([12, 12], 100, 150), ([30], 110, 0), ([8] * 5, 128, 300)]:
    cases.append(("blocks %s m %d q %d" % (sizes, m, q), core_of(sizes, m, q, 5)))
for name, core in cases:
    row = []
    for path in ("1", "0"):
        os.environ["HIPSDP_SOLVE1"] = path
        s = hb.Solver(0)
        s.load_core(core)
        best = None
        for rep in range(3):
            # (the engine prints the device cycles per phase of a one-launch solve to stderr)
            os.environ["HIPSDP_SOLVE1_PROF"] = "1" if (rep == 2 and path == "1" and name.startswith("example")) else "0"
            info = s.solve(gaptol=1e-6, feastol=1e-6)
            t = info.solve_seconds
            best = t if best is None else min(best, t)
        row.append((s.solve_path(), info.status, info.iterations, 1e3 * best / max(1, info.iterations), info.dobj))
        s.close()
    print("%-32s | one launch: path %d status %d it %2d %.3f ms/it | general: status %d it %2d %.3f ms/it | dobj diff %.1e" % (
        name, row[0][0], row[0][1], row[0][2], row[0][3], row[1][1], row[1][2], row[1][3], abs(row[0][4] - row[1][4])))
