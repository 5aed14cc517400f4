"""developer script: iteration trace of a cold and a warm-started child node of an instance's root"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, math
import bnb, sdpa_io, warm_bnb, sdpi_prepare, sdpi_call
name = sys.argv[1]; lam = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
s, solve, stats = warm_bnb.warm_node_solver(hb.lib(), 1e-6, lam)
P = sdpi_prepare.prepare(prob); P.parent_aux = None
root = solve(P)
frac = [(abs(root.y[v] - round(root.y[v])), v) for v in inst.intvars]; f, v = max(frac)
lb, ub = np.array(P.lb), np.array(P.ub); ub[v] = math.floor(root.y[v])
child = sdpi_prepare.prepare(sdpi_prepare.SdpiProblem(prob.obj, lb, ub, prob.blocks, prob.lp))
s.set_int(5, 1)
print("---- cold child"); child.parent_aux = None; solve(child); print("iterations", s.iterations())
print("---- warm child"); child.parent_aux = root.aux; solve(child); print("iterations", s.iterations())
