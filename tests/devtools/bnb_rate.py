"""developer script: bench.py's `bnb` leg alone (example_TT / example_CLS trees through SCIPsdpiSolverLoadAndSolve, no CPU figures)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
hb = bench.load_binding()
r = bench.bench_bnb(hb, cpu=False)
for k, v in r.items():
    if isinstance(v, dict):
        for lab in ("cold", "warm"):
            w = v[lab]
            print("%-12s %s: %6.1f node solves/s, %.4f ms per IPM iteration, %d nodes, %.1f iterations per node, optimum %s, one-launch solves %d" % (
                k, lab, w["node_solves_per_sec"], w["ms_per_ipm_iteration"], w["node_solves"], w["ipm_iterations_per_node"], w["optimum"], w["one_launch_engine_solves"]))
