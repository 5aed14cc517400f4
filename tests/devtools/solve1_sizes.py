"""developer script: the one-launch solve against the general path over block sizes up to its limits (sparse matrices, three
nonzeros per matrix and block): ms per interior-point iteration on both paths"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from solve1_sizes_lib import core_of


for sizes, m, q in [([10], 37, 85), ([16], 40, 40), ([24], 40, 40), ([32], 48, 40), ([48], 60, 40), ([64], 64, 64), ([12, 12, 12], 40, 40),
                    ([30, 30], 50, 20), ([20] * 8, 60, 100), ([64] * 4, 64, 200)]:
    core = core_of(sizes, m, q, 5)
    row = []
    for path in ("1", "0"):
        os.environ["HIPSDP_SOLVE1"] = path
        s = hb.Solver(0)
        s.load_core(core)
        best = None
        for rep in range(3):
            info = s.solve(gaptol=1e-6, feastol=1e-6)
            t = info.solve_seconds
            best = t if best is None else min(best, t)
        row.append((s.solve_path(), info.status, info.iterations, 1e3 * best / max(1, info.iterations), info.dobj))
        s.close()
    print("blocks %-22s m %3d q %3d | one launch: path %d status %d it %2d %.3f ms/it | general: status %d it %2d %.3f ms/it | dobj diff %.1e" % (
        sizes, m, q, row[0][0], row[0][1], row[0][2], row[0][3], row[1][1], row[1][2], row[1][3], abs(row[0][4] - row[1][4])))
