"""first node of example_small with the iteration log, for comparing two builds of the library (HIPSDP_LIB=...)"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, bnb, sdpa_io, sdpi_call, sdpi_prepare
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', 'example_small.dat-s'))
prob = bnb.instance_to_sdpi(inst)
s = sdpi_call.SdpiSolver(hb.lib())
for p in (1, 2, 3): s.set_real(p, 1e-6)
s.set_int(5, 1)
node = sdpi_prepare.SdpiProblem(prob.obj, [-10.0, -10.0, -10.0], [0.0, 1.0, 10.0], prob.blocks, prob.lp, isintegral=prob.isintegral)
rc, _, _ = s.solve(sdpi_prepare.prepare(node))
print('rc', rc, 'optimal', s.flag("IsOptimal"), 'internal', s.internal_status(), 'iters', s.iterations(), 'calls', s.sdpcalls(), 'settings', s.settings_used())
