"""developer script: the one-launch solve (csrc/solve1.hip) against the oracle, iteration by iteration.
usage: python tests/devtools/solve1_dbg.py [filter]"""
import sys, os, json, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
os.environ.setdefault("HIPSDP_SOLVE1_HIST", "1")
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref, instances, sdpa_io, sdpi_prepare

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
flt = sys.argv[1] if len(sys.argv) > 1 else ''


def case_core(case):
    blocks = [dict(n=b["n"], vars={int(k): [tuple(e) for e in v] for k, v in b["vars"].items()},
                   const=[tuple(e) for e in b["const"]]) for b in case["blocks"]]
    lp = [(l, r, {int(k): v for k, v in row.items()}) for (l, r, row) in case["lp"]]
    P = sdpi_prepare.prepare(sdpi_prepare.SdpiProblem(case["obj"], case["lb"], case["ub"], blocks, lp))
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    return ipm_ref.CoreProblem(b, blk, D, c)


def problems():
    for case in json.load(open(os.path.join(GOLDEN, "checksdpi_cases.json")))["cases"]:
        yield "golden:" + case["name"], case_core(case), 1e-6
    for name in ["example_small.dat-s", "example_TT.dat-s.gz", "example_CLS.dat-s.gz", "example_inf.dat-s", "example_tightenmatrices.dat-s"]:
        inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", name))
        D, c = sdpa_io.lp_dense(inst)
        yield "inst:" + name, ipm_ref.CoreProblem(inst.obj, sdpa_io.dense_blocks(inst), D, c), 1e-6
    for n, m in [(5, 8), (10, 20), (16, 20), (3, 2)]:
        b, A, ys, Xs, Zs = instances.planted_dense(n, m)
        yield "planted:%d,%d" % (n, m), ipm_ref.CoreProblem(b, [A]), 1e-6
    rng = np.random.default_rng(7)
    for t in range(4):
        m = int(rng.integers(3, 30)); sizes = [int(rng.integers(2, 14)) for _ in range(int(rng.integers(1, 4)))]
        q = int(rng.integers(0, 20))
        blocks = []
        ystar = rng.standard_normal(m)
        for n in sizes:
            A = np.zeros((m + 1, n, n))
            for i in range(1, m + 1):
                for _ in range(3):
                    r, c = rng.integers(0, n, 2)
                    v = rng.standard_normal()
                    A[i, r, c] += v; A[i, c, r] += v if r != c else 0.0
            Zs = rng.standard_normal((n, n)); Zs = Zs @ Zs.T + 0.5 * np.eye(n)
            A[0] = np.tensordot(ystar, A[1:], axes=(0, 0)) - Zs
            blocks.append(A)
        D = rng.standard_normal((q, m)) * (rng.random((q, m)) < 0.3)
        c = D @ ystar - rng.random(q) - 0.1
        Xs = [np.eye(n) for n in sizes]
        b = sum(np.array([np.sum(A[i] * X) for i in range(1, m + 1)]) for A, X in zip(blocks, Xs)) + (D.T @ np.ones(q) if q else 0.0)
        yield "random:%d" % t, ipm_ref.CoreProblem(b, blocks, D, c), 1e-6


names = ["it", "mu", "pinf", "dinf", "gap", "tau", "kappa", "pobj", "dobj"]
for name, core, tol in problems():
    if flt and flt not in name:
        continue
    t0 = time.time()
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=tol, feastol=tol, pabstol=10 * tol))
    tr = time.time() - t0
    s = hb.Solver(0)
    s.load_core(core)
    info = s.solve(gaptol=tol, feastol=tol, pabstol=10 * tol)
    path = s.solve_path()
    y = s.y()
    out, hist = (s.solve1_trace(256) if path else (np.zeros(64), np.zeros((0, 16))))
    s.close()
    worst = 0.0; where = None
    if path:
        for r in range(min(len(ref.history), info.iterations + 1)):
            for j in range(1, 7):
                a, bb = hist[r, j], ref.history[r][j]
                d = abs(a - bb) / max(abs(bb), 1e-30)
                if d > 1e-4 and where is None and abs(bb) > 1e-12:
                    where = (r, names[j], a, bb)
    dy = np.max(np.abs(y - ref.y)) if core.m else 0.0
    print("%-36s m %3d q %3d n %s path %d status %d/%d it %2d/%2d dobj %.9g/%.9g |dy| %.1e  %s  dev %.0f us (solve %.0f us, oracle %.0f ms) nnzA %d nnzD %d" % (
        name, core.m, core.q, [A.shape[1] for A in core.blocks], path, info.status, ref.status, info.iterations, ref.iterations,
        info.dobj, ref.dobj, dy, ("first diff %s" % (where,)) if where else "hist ok", out[43] / 100.0, info.solve_seconds * 1e6, tr * 1e3,
        int(out[40]), int(out[41])))
    if path and os.environ.get("S1_SHOWPROF"):
        print("   cycles:", " ".join("%.0f" % v for v in out[18:30]), "total", out[17])
    if where is not None and os.environ.get("S1_SHOWHIST"):
        for r in range(min(len(ref.history), info.iterations + 1)):
            print("   it %2d eng mu %.6e pinf %.3e dinf %.3e gap %.3e tau %.4e kap %.4e aa %.4f al %.4f | ref mu %.6e pinf %.3e dinf %.3e gap %.3e tau %.4e kap %.4e" % (
                r, hist[r, 1], hist[r, 2], hist[r, 3], hist[r, 4], hist[r, 5], hist[r, 6], hist[r, 9], hist[r, 10], *ref.history[r][1:7]))
