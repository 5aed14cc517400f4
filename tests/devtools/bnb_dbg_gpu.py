import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, bnb, sdpa_io, sdpi_call, sdpi_prepare, ipm_ref
name = sys.argv[1]
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
s = sdpi_call.SdpiSolver(hb.lib())
for p in (1, 2, 3): s.set_real(p, 1e-6)
s.set_int(5, 0)
def solve(P):
    rc, _, _ = s.solve(P)
    if s.flag("IsDualInfeasible"): return bnb.NodeResult('infeasible')
    if not s.flag("IsOptimal"):
        b, blk, D, c, maps = sdpi_prepare.to_core(P)
        core = ipm_ref.CoreProblem(b, blk, D, c)
        r = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-6, feastol=1e-6))
        print('FAILED node: internal', s.internal_status(), 'iters', s.iterations(), 'calls', s.sdpcalls(), '| oracle status', r.status, 'it', r.iterations, 'lb', P.lb, 'ub', P.ub)
        s.set_int(5, 1); s.solve(P); s.set_int(5, 0)
        return bnb.NodeResult('failed')
    rc, obj, y = s.dual_sol()
    return bnb.NodeResult('optimal', obj, y)
print(bnb.branch_and_bound(prob, inst.intvars, solve)[::2])
