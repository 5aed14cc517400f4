"""developer script: one random problem of tests/test_gpu_solve1.py::test_random_shapes... solved repeatedly on both paths (is a path
deterministic?).  usage: python tests/devtools/solve1_repeat.py <seed>"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
K = int(rng.integers(1, 5))
sizes = [int(rng.integers(2, 31 if K == 1 else (22 if K == 2 else 14))) for _ in range(K)]
dims = sum(n * (n + 1) // 2 for n in sizes)
m = int(rng.integers(5, max(6, min(100, dims))))
q = int(rng.integers(0, 150))
dens = float(rng.uniform(0.05, 0.5))
ystar = rng.standard_normal(m)
blocks = []
for n in sizes:
    A = np.zeros((m + 1, n, n))
    for i in range(1, m + 1):
        for _ in range(int(rng.integers(1, 4))):
            r, c = rng.integers(0, n, 2)
            v = rng.standard_normal()
            A[i, r, c] += v
            if r != c:
                A[i, c, r] += v
    Zs = rng.standard_normal((n, n)); Zs = Zs @ Zs.T + 0.5 * np.eye(n)
    A[0] = np.tensordot(ystar, A[1:], axes=(0, 0)) - Zs
    blocks.append(A)
D = rng.standard_normal((q, m)) * (rng.random((q, m)) < dens)
c = D @ ystar - rng.random(q) - 0.1
b = sum(np.array([np.trace(A[i]) for i in range(1, m + 1)]) for A in blocks) + (D.T @ np.ones(q) if q else 0.0)
core = ipm_ref.CoreProblem(b, blocks, D, c)
ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
print("sizes", sizes, "m", m, "q", q, "density %.2f" % dens, "| oracle: status", ref.status, "iterations", ref.iterations, "dobj %.12g" % ref.dobj)
for path in ("1", "0"):
    os.environ["HIPSDP_SOLVE1"] = path
    for rep in range(5):
        s = hb.Solver(0)
        s.load_core(core)
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
        print("  HIPSDP_SOLVE1=%s run %d: path %d status %d iterations %d dobj %.12g" % (path, rep, s.solve_path(), info.status, info.iterations, info.dobj))
        s.close()
