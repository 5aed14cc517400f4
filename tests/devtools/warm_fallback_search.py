"""developer script: look for WARM-started solves on which the one-launch kernel gives up numerically (the case the warm-start fallback
of csrc/ipm.hip is for).  For each seed of the fuzz family on which the kernel gives up cold, a loose general-path solve gives an
interior point; the kernel is started from it with the tight tolerances.  usage: python tests/devtools/warm_fallback_search.py [seed ...]"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
from fuzz_shapes import problem

seeds = [int(a) for a in sys.argv[1:]] or [30123, 70387, 70505, 70733, 71064, 71150]
for seed in seeds:
    core, desc = problem(seed)
    K = len(core.blocks)
    for loose in (1e-1, 1e-2, 1e-3, 1e-4):
        os.environ["HIPSDP_SOLVE1"] = "0"
        s = hb.Solver(0); s.load_core(core)
        i0 = s.solve(gaptol=loose, feastol=loose, pabstol=loose)
        y0 = s.y(); X0 = [s.X(k) for k in range(K)]; Z0 = [s.Z(k) for k in range(K)]
        x0, z0 = s.lp() if core.q > 0 else (None, None)
        s.close()
        res = {}
        for mode in ("kernel", "default", "general"):
            os.environ["HIPSDP_SOLVE1"] = "0" if mode == "general" else "1"
            if mode == "kernel":
                os.environ["HIPSDP_SOLVE1_NO_FALLBACK"] = "1"
            else:
                os.environ.pop("HIPSDP_SOLVE1_NO_FALLBACK", None)
            s = hb.Solver(0); s.load_core(core)
            s.set_start(y0, X0, Z0, x0, z0)
            info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
            res[mode] = (s.solve_path(), info.status, info.iterations, info.warm_started, info.dobj, s.y())
            s.close()
        k, d, g = res["kernel"], res["default"], res["general"]
        print("seed %d %s loose %.0e (%d it): kernel %s default %s general %s same-as-general %s" % (
            seed, desc, loose, i0.iterations, k[:5], d[:5], g[:5], np.array_equal(d[5], g[5])), flush=True)
