"""developer script: one shape of the fuzz family (tests/harness/fuzz_shapes.py) iteration by iteration - the one-launch kernel's
history (mu, residuals, steps, forced pivots of the factorization of M, residual of the linearised primal equation) beside the oracle's
and the general path's outcome.  usage: python tests/devtools/solve1_fuzz_one.py seed [seed ...]"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
os.environ["HIPSDP_SOLVE1_HIST"] = "1"
os.environ["HIPSDP_SOLVE1_NO_FALLBACK"] = "1"
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import ipm_ref
from fuzz_shapes import problem

for seed in [int(a) for a in sys.argv[1:]]:
    core, desc = problem(seed)
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
    print("seed %d %s: oracle status %d it %d dobj %.10g" % (seed, desc, ref.status, ref.iterations, ref.dobj))
    for path in ("1", "0"):
        os.environ["HIPSDP_SOLVE1"] = path
        s = hb.Solver(0); s.load_core(core)
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
        print("  HIPSDP_SOLVE1=%s: path %d status %d it %d dobj %.10g" % (path, s.solve_path(), info.status, info.iterations, info.dobj))
        if path == "1" and s.solve_path() == 1:
            out, hist = s.solve1_trace(40)
            if info.status == 5:
                print("   the kernel gave up at csrc/solve1_body.h:%d" % int(out[44]))
            print("   it        mu      pinf      dinf       gap       tau     kappa | pred.step   step      dtau   lin.res   forced      |dy|  || oracle: mu pinf dinf gap tau")
            for r in hist[:info.iterations + 1]:
                it = int(r[0])
                o = ref.history[it] if it < len(ref.history) else None
                print("   %2d %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e | %8.4f %8.4f %9.2e %9.2e %8.0f %9.2e  || %s" % (
                    it, r[1], r[2], r[3], r[4], r[5], r[6], r[9], r[10], r[11], r[12], r[13], r[14],
                    " ".join("%9.2e" % v for v in o[1:6]) if o is not None else ""))
        s.close()
