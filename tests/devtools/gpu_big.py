"""developer script: one large planted instance on the GPU, objective vs the planted optimum"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np
import instances
n, m = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t = time.time(); b, A, ys, Xs, Zs = instances.planted_dense(n, m); print('gen %.1fs' % (time.time() - t), flush=True)
s = hb.Solver(0)
s.set_shape(m, [n], 0); s.set_obj(b)
t = time.time(); s.set_block_dense(0, A); print('upload %.2fs' % (time.time() - t), flush=True)
for r in range(reps):
    t = time.time(); info = s.solve(gaptol=1e-5, feastol=1e-5, verbose=(r == 0)); t = time.time() - t
    y = s.y()
    print('solve %d: status %d it %d dobj %.10g pobj %.10g opt %.10g |y-y*| %.2e pinf %.1e dabs %.1e  wall %.3fs  schur %.3fs (%.1f TF algorithmic) cholfail %d' % (
        r, info.status, info.iterations, info.dobj, info.pobj, b @ ys, np.abs(y - ys).max(), info.pinf, info.dabs, t, info.schur_seconds,
        info.schur_flops / max(info.schur_seconds, 1e-9) / 1e12, info.chol_fail), flush=True)
