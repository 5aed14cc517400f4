"""time stamps inside k_lmin_tiny (build of eig.hip with -DEIG_TIMING, HIPSDP_LIB=...): products / tridiagonalisation / setup / multisection"""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness")); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, bnb, sdpa_io, sdpi_call
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', 'example_TT.dat-s.gz'))
prob = bnb.instance_to_sdpi(inst)
s = sdpi_call.SdpiSolver(hb.lib())
for p in (1, 2, 3): s.set_real(p, 1e-6)
import sdpi_prepare
rc, _, _ = s.solve(sdpi_prepare.prepare(prob))
out = (C.c_longlong * 8)()
fn = hb.lib().hipsdp_debug_lm_timing
fn.argtypes = [C.POINTER(C.c_longlong)]
assert fn(out) == 0
t = [out[i] for i in range(5)]
print("iterations", s.iterations(), "stamps (us): products %.2f, tridiagonalisation %.2f, interval %.2f, multisection %.2f" % tuple((t[i + 1] - t[i]) / 100.0 for i in range(4)))
