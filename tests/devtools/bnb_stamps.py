"""developer script: the one-launch kernel's own section times (S1_STAMP sums, HIPSDP_SOLVE1_PROF=1) summed over every node of a tree
solved through SCIPsdpiSolverLoadAndSolve - what an iteration costs at the NODES of a tree (fixed variables merged into the constant
matrix, tightened bounds), not at its root loaded as a dense core.
usage: python tests/devtools/bnb_stamps.py [instance] [maxnodes]"""
import sys, os, importlib.util
os.environ["HIPSDP_SOLVE1_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import ctypes as C
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bnb, sdpa_io, sdpi_call
name = sys.argv[1] if len(sys.argv) > 1 else "example_TT.dat-s.gz"
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
lib = hb.lib()
s = sdpi_call.SdpiSolver(lib)
for p in (1, 2, 3): s.set_real(p, 1e-6)
tot = np.zeros(64); its = 0; nodes = 0
lib.hipsdp_solve1_trace.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
def solve(P):
    global its, nodes
    s.solve(P)
    out = np.zeros(64)
    ptr = lib.SCIPsdpiSolverGetSolverPointer(s.h)
    if lib.hipsdp_solve1_trace(C.c_void_p(ptr), out.ctypes.data_as(C.POINTER(C.c_double)), 0, None) == 0 and out[17] > 0:
        tot[:] += out; nodes += 1
        it = C.c_int(0); lib.SCIPsdpiSolverGetIterations(s.h, C.byref(it)); its += it.value
    if s.flag("IsDualInfeasible"): return bnb.NodeResult('infeasible')
    if not s.flag("IsOptimal"): return bnb.NodeResult('failed')
    rc, obj, y = s.dual_sol()
    return bnb.NodeResult('optimal', obj, y)
r = bnb.branch_and_bound(prob, inst.intvars, solve, maxnodes=int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
names = {0: "setup", 1: "start", 2: "residuals", 3: "inverse factors", 4: "Zinv", 5: "Schur", 6: "chol M + predictor rhs + solves", 7: "B, S0, dtau",
         12: "dZ", 13: "X dZ", 15: "dX (predictor)", 8: "E + predictor step length", 16: "  steplen products a", 17: "  steplen products b",
         18: "  steplen eigenvalues", 9: "corrector rhs + solve", 10: "corrector dZ, dX", 11: "corrector step length + update",
         20: "  (factorization of M, wavefront 0)", 21: "  (solve behind it)"}
print("%d node solves in the kernel, %d iterations, kernel cycles per iteration %.0f (setup included: %.0f per solve)" % (
    nodes, its, (tot[17] - tot[18]) / max(1, its), tot[18] / max(1, nodes)))
for i in range(22):
    if tot[18 + i] != 0:
        print("  stamp %2d %-40s %10.0f cycles per %s" % (i, names.get(i, ""), tot[18 + i] / (nodes if i in (0, 1) else max(1, its)), "solve" if i in (0, 1) else "iteration"))
