"""developer script: the B&B tree of example_MkP (m = 105, n = 15, 240 LP rows) through SCIPsdpiSolverLoadAndSolve, one-launch kernel
on and off"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'harness')]
import bench
hb = bench.load_binding()
import bnb, sdpa_io, warm_bnb
inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", "example_MkP.dat-s.gz"))
prob = bnb.instance_to_sdpi(inst)
for on in ("1", "0"):
    os.environ["HIPSDP_SOLVE1"] = on
    s, solve, stats = warm_bnb.warm_node_solver(hb.lib(), 1e-6, 0.0)
    t0 = time.perf_counter()
    best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, solve)
    wall = time.perf_counter() - t0
    s.free()
    print("HIPSDP_SOLVE1=%s: optimum %s, %d nodes, %d node solves, %.1f node solves/s, %.4f ms per iteration, %.1f iterations per node, tree %.1f s" % (
        on, best, nodes, stats["calls"], stats["calls"] / stats["wall"], 1e3 * stats["time"] / max(1, stats["iters"]), stats["iters"] / max(1, stats["calls"]), wall))
