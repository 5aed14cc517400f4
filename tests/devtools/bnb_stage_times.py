"""developer script: wall time of the marshalling stages of SCIPsdpiSolverLoadAndSolve over the first nodes of a tree (HIPSDP_STAGE_TIMES=1)"""
import sys, os, importlib.util
os.environ["HIPSDP_STAGE_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bnb, sdpa_io, sdpi_call, time
name = sys.argv[1] if len(sys.argv) > 1 else "example_TT.dat-s.gz"
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
s = sdpi_call.SdpiSolver(hb.lib())
for p in (1, 2, 3): s.set_real(p, 1e-6)
st = dict(n=0, solve=0.0, get=0.0)
def solve(P):
    t0 = time.perf_counter(); s.solve(P); t1 = time.perf_counter()
    st['n'] += 1; st['solve'] += t1 - t0
    if s.flag("IsDualInfeasible"): return bnb.NodeResult('infeasible')
    if not s.flag("IsOptimal"): return bnb.NodeResult('failed')
    rc, obj, y = s.dual_sol(); st['get'] += time.perf_counter() - t1
    return bnb.NodeResult('optimal', obj, y)
r = bnb.branch_and_bound(prob, inst.intvars, solve, maxnodes=int(sys.argv[2]) if len(sys.argv) > 2 else 60)
print("nodes", st['n'], "LoadAndSolve %.1f us per node, flags + dual_sol %.1f us per node, engine opttime" % (1e6 * st['solve'] / st['n'], 1e6 * st['get'] / st['n']))
