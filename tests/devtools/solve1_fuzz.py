"""developer script: random shapes over everything the one-launch kernel is offered, each solved twice by the kernel and once by the
general path; prints every disagreement (status, iteration count, objective, run-to-run differences).
usage: python tests/devtools/solve1_fuzz.py [first seed] [count]"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300


from fuzz_shapes import problem


def solve(core, path):
    os.environ["HIPSDP_SOLVE1"] = path
    os.environ["HIPSDP_SOLVE1_NO_FALLBACK"] = "1"          # the kernel's own verdict
    s = hb.Solver(0)
    s.load_core(core)
    info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
    r = (s.solve_path(), info.status, info.iterations, info.dobj)
    s.close()
    return r


bad = taken = 0
for seed in range(first, first + count):
    core, tag = problem(seed)
    a, a2, g = solve(core, "1"), solve(core, "1"), solve(core, "0")
    taken += a[0]
    msg = None
    if a != a2:
        msg = "RUN-TO-RUN %s / %s" % (a, a2)
    elif a[1] >= 4 and g[1] >= 4:
        pass
    elif a[1] != g[1] or a[2] != g[2] or (a[1] == 0 and abs(a[3] - g[3]) > 1e-7 * (1 + abs(g[3]))):
        msg = "PATHS DIFFER one-launch %s general %s" % (a, g)
    if msg:
        bad += 1
        print("seed %d %s: %s" % (seed, tag, msg))
print("%d problems, %d taken by the kernel, %d disagreements" % (count, taken, bad))
