"""developer script: one saved node problem (npz written by bnb_ab_paths.py with AB_DEEP=1) through the one-launch kernel with its
diagnostics: residual of the linearised primal equation, forced pivots, |dy|, |h| per iteration"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
os.environ["HIPSDP_SOLVE1_HIST"] = "1"
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref
d = np.load(sys.argv[1])
blk = [d[k] for k in sorted(d.files) if k.startswith('A')]
core = ipm_ref.CoreProblem(d['b'], blk, d['D'], d['c'])
ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
s = hb.Solver(0); s.load_core(core)
info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5, maxiter=int(os.environ.get("MAXIT", "30")))
out, hist = s.solve1_trace(256)
print("engine status %d it %d | oracle status %d it %d" % (info.status, info.iterations, ref.status, ref.iterations))
nb = np.linalg.norm(core.b)
for r in range(info.iterations + 1):
    rp = hist[r, 2] * hist[r, 5] * (1 + nb)
    print("it %2d mu %.3e pinf %.3e |rp| %.3e tau %.3e al %.4f | err %.3e forced %d zeroed %d |dy| %.3e solve resid %.3e | ref pinf %s" % (
        r, hist[r, 1], hist[r, 2], rp, hist[r, 5], hist[r, 10], hist[r, 12], int(hist[r, 13]) % 65536, int(hist[r, 13]) // 65536, hist[r, 14], hist[r, 15],
        ("%.3e" % ref.history[r][2]) if r < len(ref.history) else "-"))
