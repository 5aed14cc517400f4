"""developer script: the bench instance (planted dense block, as bench.py builds it) at a few sizes through the general path; prints status,
iterations, objective and the first entries of y as hex (HIPSDP_LIB selects the library: two builds compared with diff)."""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
hb = bench.load_binding()
for n, m in [(64, 100), (130, 200), (300, 400), (500, 1000)]:
    s = hb.Solver(0)
    s.set_shape(m, [n], 0)
    Xs, Zs, ys = bench.planted_pair(n, m, 1)
    b = s.gen_planted(n, m, 1, Xs, Zs, ys)
    info = s.solve(gaptol=1e-5, feastol=1e-5)
    y = s.y()
    print("n %d m %d status %d iterations %d dobj %s y %s" % (n, m, info.status, info.iterations, float(info.dobj).hex(), y.tobytes().hex()[:96]))
    s.close()
