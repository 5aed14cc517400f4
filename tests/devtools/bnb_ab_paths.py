"""developer script: B&B on one instance, every node solved by the one-launch kernel AND by the general path; prints the nodes
whose outcomes or iteration counts differ.  usage: python tests/devtools/bnb_ab_paths.py example_TT.dat-s.gz [maxnodes]"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bnb, sdpa_io, sdpi_call
name = sys.argv[1]; maxnodes = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
sa = sdpi_call.SdpiSolver(hb.lib()); sb = sdpi_call.SdpiSolver(hb.lib())
for s in (sa, sb):
    for p in (1, 2, 3): s.set_real(p, 1e-6)
tot = dict(a=0, b=0, ta=0.0, tb=0.0, n=0, diff=0)
def outcome(s):
    if s.flag("IsDualInfeasible"): return 'infeasible', None
    if not s.flag("IsOptimal"): return 'failed', None
    rc, obj, y = s.dual_sol(); return 'optimal', (obj, y)

def deep(P):
    """the first differing node at engine level: one-launch history beside the oracle's"""
    import numpy as np, ipm_ref, sdpi_prepare
    b, blk, D, c, maps = sdpi_prepare.to_core(P)
    core = ipm_ref.CoreProblem(b, blk, D, c)
    np.savez(os.path.join(ROOT, 'gpurun_out', 'ab_node.npz'), b=b, D=D, c=c, **{'A%d' % k: A for k, A in enumerate(blk)})
    ref = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6, pabstol=1e-5))
    for path in ("1", "0"):
        os.environ["HIPSDP_SOLVE1"] = path; os.environ["HIPSDP_SOLVE1_HIST"] = "1"
        s = hb.Solver(0); s.load_core(core)
        info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
        print("engine path %s: status %d it %d (oracle status %d it %d)" % (s.solve_path(), info.status, info.iterations, ref.status, ref.iterations))
        if s.solve_path():
            out, hist = s.solve1_trace(256)
            for r in range(min(len(ref.history), info.iterations + 1)):
                print("   it %2d eng mu %.6e pinf %.3e dinf %.3e gap %.3e tau %.4e kap %.4e aa %.4f al %.6f | ref mu %.6e pinf %.3e dinf %.3e gap %.3e tau %.4e kap %.4e" % (
                    r, hist[r, 1], hist[r, 2], hist[r, 3], hist[r, 4], hist[r, 5], hist[r, 6], hist[r, 9], hist[r, 10], *ref.history[r][1:7]))
        s.close()

def solve(P):
    os.environ["HIPSDP_SOLVE1"] = "1"
    t0 = time.perf_counter(); sa.solve(P); tot['ta'] += time.perf_counter() - t0
    ia = sa.iterations(); oa = outcome(sa)
    os.environ["HIPSDP_SOLVE1"] = "0"
    t0 = time.perf_counter(); sb.solve(P); tot['tb'] += time.perf_counter() - t0
    ib = sb.iterations(); ob = outcome(sb)
    tot['a'] += ia; tot['b'] += ib; tot['n'] += 1
    if oa[0] != ob[0] or abs(ia - ib) > 1 or (oa[0] == 'optimal' and abs(oa[1][0] - ob[1][0]) > 1e-5 * (1 + abs(ob[1][0]))):
        tot['diff'] += 1
        if tot['diff'] == 1 and os.environ.get("AB_DEEP"):
            deep(P)
        if tot['diff'] <= 40:
            print("node %d: one-launch %s it %d obj %s | general %s it %d obj %s" % (tot['n'], oa[0], ia, oa[1][0] if oa[1] else None, ob[0], ib, ob[1][0] if ob[1] else None))
    if oa[0] != 'optimal': return bnb.NodeResult(oa[0])
    return bnb.NodeResult('optimal', oa[1][0], oa[1][1])
r = bnb.branch_and_bound(prob, inst.intvars, solve, maxnodes=maxnodes)
print(name, 'best', r[0], 'nodes', r[2], 'iterations one-launch %d general %d' % (tot['a'], tot['b']), 'differing nodes', tot['diff'],
      'wall one-launch %.3f s general %.3f s' % (tot['ta'], tot['tb']))
