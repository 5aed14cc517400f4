#!/usr/bin/env python3
"""leak_check.py - developer tool: device memory must return to its starting level after many create / shape / solve / free
cycles (engine and solver-interface level)"""
import os, sys, importlib.util, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness")); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import numpy as np, ipm_ref, instances, sdpi_prepare, sdpi_call, json
hip = C.CDLL("libamdhip64.so")
def free_mem():
    f = C.c_size_t(); t = C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value
s0 = hb.Solver(0); s0.close()            # context creation
base = free_mem()
for rep in range(60):
    n, m = [(8, 12), (40, 70), (70, 130), (20, 200)][rep % 4]
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    core = ipm_ref.CoreProblem(b, [A])
    s = hb.Solver(0); s.load_core(core); info = s.solve(gaptol=1e-6, feastol=1e-6)
    assert info.status == 0
    if rep % 2: s.load_core(core); s.solve(gaptol=1e-5, feastol=1e-5)      # re-shape the same handle
    s.close()
mid = free_mem()
cases = json.load(open(os.path.join(ROOT, "tests", "golden", "checksdpi_cases.json")))["cases"]
import test_gpu_sdpi as T
for rep in range(30):
    case = cases[rep % len(cases)]
    P = sdpi_prepare.prepare(T.build(case))
    if P.status != 'ok': continue
    s = T.new_solver(hb); s.solve(P); s.solve(P); s.free()
end = free_mem()
print("free device memory: start %.1f MB, after engine cycles %+.1f MB, after interface cycles %+.1f MB" % (base / 1e6, (mid - base) / 1e6, (end - base) / 1e6))
print("host memory of the interface (counting allocator):", hb.lib().hipsdp_compat_mem_used() if hasattr(hb.lib(), "hipsdp_compat_mem_used") else "n/a")
# the runtime keeps ~100 MB of pools / code objects after first use; what must not happen is growth with the number of cycles
sys.exit(0 if (mid - end) < 32e6 and (base - mid) < 512e6 else 1)
