"""developer script: B&B on one instance over the HIP backend (for profiling)"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bnb, sdpa_io, sdpi_call
name = sys.argv[1]; maxnodes = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
s = sdpi_call.SdpiSolver(hb.lib())
for p in (1, 2, 3): s.set_real(p, 1e-6)
it = [0]
def solve(P):
    rc, _, _ = s.solve(P); it[0] += s.iterations()
    if s.flag("IsDualInfeasible"): return bnb.NodeResult('infeasible')
    if not s.flag("IsOptimal"): return bnb.NodeResult('failed')
    rc, obj, y = s.dual_sol(); return bnb.NodeResult('optimal', obj, y)
t = time.time(); r = bnb.branch_and_bound(prob, inst.intvars, solve, maxnodes=maxnodes); t = time.time() - t
print(name, 'best', r[0], 'nodes', r[2], 'iterations', it[0], 'time %.2f' % t)
