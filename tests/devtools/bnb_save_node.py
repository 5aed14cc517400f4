"""developer script: walk a tree through SCIPsdpiSolverLoadAndSolve and save the core problem of node number N (as the engine sees it:
fixed variables merged into the constant matrix, tightened bounds as LP rows) to an npz file for solve1_waves.py / solve1_node.py;
prints the entry counts of its matrices.  usage: python tests/devtools/bnb_save_node.py instance N outfile"""
import sys, os, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'harness'), os.path.join(ROOT, 'tests')]
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bnb, sdpa_io, sdpi_call, sdpi_prepare
name, N, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
s = sdpi_call.SdpiSolver(hb.lib())
for p in (1, 2, 3): s.set_real(p, 1e-6)
cnt = [0]
def solve(P):
    cnt[0] += 1
    if cnt[0] == N:
        b, blk, D, c, maps = sdpi_prepare.to_core(P)
        np.savez(out, b=b, D=D, c=c, **{'A%d' % k: A for k, A in enumerate(blk)})
        for k, A in enumerate(blk):
            ent = [(int(np.count_nonzero(A[i])), int(np.count_nonzero(np.abs(A[i]).sum(axis=1)))) for i in range(A.shape[0])]
            print("node %d block %d: n %d, m %d; (entries, non-empty rows) of the constant matrix %s; of the variables: %s" % (
                N, k, A.shape[1], A.shape[0] - 1, ent[0], sorted(set(ent[1:]))))
        print("LP rows %d, nonzeros %d" % (D.shape[0], int(np.count_nonzero(D))))
    s.solve(P)
    if s.flag("IsDualInfeasible"): return bnb.NodeResult('infeasible')
    if not s.flag("IsOptimal"): return bnb.NodeResult('failed')
    rc, obj, y = s.dual_sol()
    return bnb.NodeResult('optimal', obj, y)
bnb.branch_and_bound(prob, inst.intvars, solve, maxnodes=N + 1)
