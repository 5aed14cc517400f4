"""developer script: the GENERAL path (HIPSDP_SOLVE1=0) on the shapes of solve1_fuzz.py, on sized problems with blocks of 17-48 rows and
on the root of example_CLS; one line per problem - status, iterations, objective and y as hex - so that two builds of the library can be
compared bit for bit with diff (HIPSDP_LIB selects the library).
usage: python tests/devtools/general_dump.py first count outfile"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'devtools')]
import numpy as np
os.environ["HIPSDP_SOLVE1"] = "0"
first, count, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import fuzz_shapes as fz
import ipm_ref, sdpa_io
import test_gpu_solve1


def line(tag, core):
    s = hb.Solver(0)
    s.load_core(core)
    info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
    y = s.y()
    r = "%s %d %d %d %s %s" % (tag, s.solve_path(), info.status, info.iterations, float(info.dobj).hex(), y.tobytes().hex()[:96])
    s.close()
    return r


lines = []
for seed in range(first, first + count):
    core, tag = fz.problem(seed)
    lines.append(line("fuzz%d" % seed, core))
for sizes, m, q in [([17], 30, 10), ([24], 40, 40), ([32], 48, 40), ([33], 40, 0), ([40], 30, 10), ([43], 33, 0), ([48], 60, 20), ([30, 30], 50, 20),
                    ([20, 45], 40, 5), ([48, 17, 5], 64, 12)]:
    for sd in (5, 6):
        lines.append(line("sized%s_%d_%d_%d" % ("x".join(map(str, sizes)), m, q, sd), test_gpu_solve1.sized_sparse_core(sizes, m, q, sd)))
for name in ("example_CLS.dat-s.gz", "example_MkP.dat-s.gz", "example_TT.dat-s.gz"):
    inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
    D, c = sdpa_io.lp_dense(inst)
    lines.append(line(name, ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)))
open(out, "w").write("\n".join(lines) + "\n")
print("%d problems on the general path" % len(lines))
