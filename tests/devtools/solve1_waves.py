"""developer script: which wavefront each phase of the one-launch solve waits for.  HIPSDP_SOLVE1_PROF=2 makes every wavefront note
the clock when it reaches each barrier of iteration 3; printed per barrier: the time from the previous barrier's release to the
arrival of each of the eight wavefronts (cycles).
usage: python tests/devtools/solve1_waves.py [instance]"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
os.environ["HIPSDP_SOLVE1_HIST"] = "1"
os.environ["HIPSDP_SOLVE1_PROF"] = "2"
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref, sdpa_io

name = sys.argv[1] if len(sys.argv) > 1 else "example_TT.dat-s.gz"
if name.startswith("random:"):
    # random:<n,n,...>:<m>:<q>  - sparse variable matrices (three nonzeros each), dense constant matrices, LP rows of density 0.3
    import test_gpu_solve1
    _, ns, m_, q_ = name.split(":")
    core = test_gpu_solve1.sized_sparse_core([int(v) for v in ns.split(",")], int(m_), int(q_), 5)
elif name.endswith(".npz"):
    # a node problem saved by tests/devtools/bnb_save_node.py
    d = np.load(name)
    core = ipm_ref.CoreProblem(d['b'], [d[k] for k in sorted(d.files) if k.startswith('A')], d['D'], d['c'])
else:
    inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
    D, c = sdpa_io.lp_dense(inst)
    core = ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c)
s = hb.Solver(0)
s.load_core(core)
info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
assert s.solve_path() == 1
out, hist = s.solve1_trace(256)
s.close()
st = hist.reshape(-1)[2048:2048 + 480].reshape(60, 8)
prev = None
tot = 0.0
print("barrier  wait-for  " + " ".join("w%d" % w for w in range(8)))
for b in range(60):
    if st[b].max() == 0:
        break
    if prev is not None:
        d = st[b] - prev
        print("%3d  %7.0f   %s" % (b, d.max(), " ".join("%6.0f" % v for v in d)))
        tot += d.max()
    prev = st[b].max()
print("sum of the phases of iteration 3: %.0f cycles; %d iterations" % (tot, info.iterations))
su = hist.reshape(-1)[2048 + 480:2048 + 480 + 16]
print("setup stamps (cycles since kernel start):", " ".join("%.0f" % v for v in su))
# per-solve sums of the stamped sections (S1_STAMP ids 1 .. 18: cycles of thread 0 between two stamps, summed over the iterations;
# 19: Sturm multisection of the first step-length task, 20: factorization of M, 21: the two solves behind it)
pr = out[18:40]
print("stamp sums per iteration (cycles):", " ".join("%d:%.0f" % (i, pr[i] / max(1, info.iterations)) for i in range(22) if pr[i] != 0))
print("kernel cycles %.0f, wall %.1f us, iterations %d" % (out[17], out[43] * 1e-2, info.iterations))
