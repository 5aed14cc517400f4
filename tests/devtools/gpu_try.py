"""developer smoke script: unit kernels vs numpy, solves vs the oracle (run on the GPU box)"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np
import scipy.linalg as sla
import ipm_ref, instances, sdpa_io

rng = np.random.default_rng(1)
print('devices', hb.device_count())

def rel(a, b): return np.abs(a - b).max() / max(1e-300, np.abs(b).max())

# --- units
for n in [1, 2, 7, 64, 65, 130, 300]:
    G = rng.standard_normal((n, n)); S = G @ G.T + n * np.eye(n)
    L, fail = hb.potrf(S)
    print('potrf', n, fail, rel(L, np.linalg.cholesky(S)))
    Li = hb.trtri(S)
    print('trtri', n, rel(Li, np.linalg.inv(np.linalg.cholesky(S))))
    r = rng.standard_normal((2, n))
    xs = hb.potrs(S, r)
    print('potrs', n, rel(xs, np.linalg.solve(S, r.T).T))
    W = G + G.T
    th, rs = hb.lambda_min(W, 0)
    print('lmin ', n, th, rs, np.linalg.eigvalsh(W)[0])
    if n <= 130:
        lam, V = hb.syev(W)
        ev, U = np.linalg.eigh(W)
        print('syev ', n, rel(lam, ev), np.abs(V @ W @ V.T - np.diag(lam)).max(), np.abs(V @ V.T - np.eye(n)).max())
for (R, E) in [(5, 100), (37, 10000), (300, 40001), (1001, 2500)]:
    A = rng.standard_normal((R, E)); V = rng.standard_normal((3, E)); c = rng.standard_normal(R)
    print('gemv_n', R, E, rel(hb.gemv_n(A, V), V @ A.T), 'gemv_t', rel(hb.gemv_t(A, c), c @ A))
for (m1, n) in [(4, 3), (9, 5), (41, 20), (101, 50), (150, 64)]:
    A = rng.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    G = rng.standard_normal((n, n)); X = G @ G.T + np.eye(n)
    G = rng.standard_normal((n, n)); Zi = np.linalg.inv(G @ G.T + np.eye(n))
    Mx = hb.schur_dense(A, X, Zi)
    ref = ipm_ref.schur_block(A, X, Zi)
    print('schur', m1, n, rel(Mx, ref), 'chunked', rel(hb.schur_dense(A, X, Zi, ws_gbytes=16.0 * n * n * 3 / 1e9), ref))

# --- solves
def solve_both(name, prob, **kw):
    P = ipm_ref.Params(gaptol=kw.get('gaptol', 1e-6), feastol=kw.get('feastol', 1e-6))
    t = time.time(); r = ipm_ref.hsd_solve(prob, P); tc = time.time() - t
    s = hb.Solver(0); s.load_core(prob)
    t = time.time(); info = s.solve(gaptol=P.gaptol, feastol=P.feastol, verbose=kw.get('verbose', 0)); tg = time.time() - t
    y = s.y()
    print('%-22s oracle st %d it %2d dobj %.9g | gpu st %d it %2d dobj %.9g pobj %.9g pinf %.1e dabs %.1e cholfail %d | dy %.2e  t_cpu %.3f t_gpu %.3f schur %.3fs' % (
        name, r.status, r.iterations, r.dobj, info.status, info.iterations, info.dobj, info.pobj, info.pinf, info.dabs, info.chol_fail,
        np.abs(y - r.y).max() if len(y) else 0.0, tc, tg, info.schur_seconds))
    s.close()
    return r, info

CP = ipm_ref.CoreProblem
solve_both('t1', CP([-3, -1], [], [[-2, -1], [-1, -3], [1, 0], [0, 1]], [-10, -15, 0, 0]))
solve_both('t2', CP([-3, -1], [], [[-2, -1], [-1, -3]], [-10, -15]))
solve_both('t3', CP([10, 15], [], [[2, 1], [-2, -1], [1, 3], [-1, -3], [1, 0], [0, 1]], [3, -3, 1, -1, 0, 0]))
solve_both('t4', CP([-1, -1], [], [[-1, 1], [1, -1]], [0, 1]))
A = np.zeros((3, 2, 2)); A[0] = [[0, -1], [-1, 0]]; A[1, 0, 0] = 1; A[2, 1, 1] = 0.75
Db = [[1, 0], [-1, 0], [0, 1], [0, -1]]; cb = [-1, -1, -1, -1]
solve_both('t9', CP([-1, 0], [A], Db, cb))
A = np.zeros((3, 2, 2)); A[1, 0, 0] = 1; A[2, 1, 1] = 1
solve_both('t10', CP([-1, -1], [A], Db, cb))
A = np.zeros((2, 2, 2)); A[0] = [[1, 2], [2, 4]]; A[1] = np.eye(2)
solve_both('t11', CP([1], [A]))
inst_dir = os.path.join(ROOT, 'tests', 'golden', 'instances')
for name in sorted(os.listdir(inst_dir)) if os.path.isdir(inst_dir) else []:
    inst = sdpa_io.read_sdpa(os.path.join(inst_dir, name))
    D, c = sdpa_io.lp_dense(inst)
    solve_both(name, CP(inst.obj, sdpa_io.dense_blocks(inst), D, c))
for (n, m) in [(5, 8), (20, 40), (50, 100), (100, 200), (200, 400)]:
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    r, info = solve_both('planted %d/%d' % (n, m), CP(b, [A]), gaptol=1e-5, feastol=1e-5)
    print('   optimum', b @ ys)
