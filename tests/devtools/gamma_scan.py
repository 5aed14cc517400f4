"""step-length factor gamma on the bench instance (n = 500, m = 1000) and on T1: iterations and time per solve (developer tool)"""
import sys, os, time, importlib.util
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bench
for n, m in ((500, 1000), (1000, 2000)):
    s = hb.Solver(0)
    s.set_shape(m, [n], 0)
    Xs, Zs, ys = bench.planted_pair(n, m, 20240)
    b = s.gen_planted(n, m, 20240, Xs, Zs, ys)
    opt = float(b @ ys)
    for g, ls in ((0.98, 0), (0.99, 0), (0.995, 0), (0.999, 0), (0.98, 40), (0.999, 40), (0.9999, 40)):
        s.solve(gaptol=1e-5, feastol=1e-5, gamma=g, lanczos_steps=ls)
        t0 = time.time()
        info = s.solve(gaptol=1e-5, feastol=1e-5, gamma=g, lanczos_steps=ls)
        dt = time.time() - t0
        print("n %d gamma %.4f lanczos steps %d: status %d iterations %d %.1f ms objective error %.2e" % (n, g, ls, info.status, info.iterations, 1e3 * dt, abs(info.dobj - opt)), flush=True)
    s.close()
