"""developer script: the one-launch kernel on the shapes of solve1_fuzz.py, every solve twice; writes one line per seed - path, status,
iterations, objective and y as hex - so that two builds of the library (release / -DS1_DEBUG) can be compared bit for bit with diff.
usage: python tests/devtools/solve1_dump.py first count outfile"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'devtools')]
import ctypes as C
import numpy as np
os.environ["HIPSDP_SOLVE1"] = "1"
first, count, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import fuzz_shapes as fz
lines = []
nonrep = 0
for seed in range(first, first + count):
    core, tag = fz.problem(seed)
    res = []
    for rep in range(2):
        s = hb.Solver(0)
        s.load_core(core)
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
        y = s.y()
        res.append("%d %d %d %s %s" % (s.solve_path(), info.status, info.iterations, float(info.dobj).hex(), y.tobytes().hex()[:64]))
        s.close()
    if res[0] != res[1]:
        nonrep += 1
    lines.append("%d %s | %s" % (seed, res[0], "same" if res[0] == res[1] else res[1]))
open(out, "w").write("\n".join(lines) + "\n")
cnt = (C.c_uint * 2)()
dbg = hb.lib().hipsdp_solve1_debug_counts(cnt)
print("%d shapes, %d not reproduced by a second solve; debug build %d: %d values declared wave-uniform were not, %d solves in the kernel" %
      (count, nonrep, dbg, cnt[0], cnt[1]))
