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


def problem(seed):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(1, 6))
    sizes = [int(rng.integers(1, 31 if K == 1 else (22 if K == 2 else 14))) for _ in range(K)]
    dims = sum(n * (n + 1) // 2 for n in sizes)
    m = int(rng.integers(1, max(2, min(110, dims))))
    q = int(rng.integers(0, 200))
    dens = float(rng.uniform(0.02, 0.6))
    ystar = rng.standard_normal(m)
    blocks = []
    for n in sizes:
        A = np.zeros((m + 1, n, n))
        for i in range(1, m + 1):
            for _ in range(int(rng.integers(0, 5))):
                r, c = rng.integers(0, n, 2)
                v = rng.standard_normal()
                A[i, r, c] += v
                if r != c:
                    A[i, c, r] += v
        Zs = rng.standard_normal((n, n)); Zs = Zs @ Zs.T + 0.5 * np.eye(n)
        A[0] = np.tensordot(ystar, A[1:], axes=(0, 0)) - Zs
        if rng.random() < 0.3:
            A[0] = np.diag(np.diag(A[0])) - 0.0          # sparse constant matrix in some blocks
        blocks.append(A)
    D = rng.standard_normal((q, m)) * (rng.random((q, m)) < dens)
    c = D @ ystar - rng.random(q) - 0.1
    b = sum(np.array([np.trace(A[i]) for i in range(1, m + 1)]) for A in blocks) + (D.T @ np.ones(q) if q else 0.0)
    return ipm_ref.CoreProblem(b, blocks, D, c), "sizes %s m %d q %d density %.2f" % (sizes, m, q, dens)


def solve(core, path):
    os.environ["HIPSDP_SOLVE1"] = path
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
    elif a[1] != g[1] or abs(a[2] - g[2]) > (1 if a[1] == 0 else 2) or (a[1] == 0 and abs(a[3] - g[3]) > 1e-7 * (1 + abs(g[3]))):
        msg = "PATHS DIFFER one-launch %s general %s" % (a, g)
    if msg:
        bad += 1
        print("seed %d %s: %s" % (seed, tag, msg))
print("%d problems, %d taken by the kernel, %d disagreements" % (count, taken, bad))
