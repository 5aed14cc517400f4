"""developer script: B&B with warm-started nodes (tests/warm_bnb.py), iteration counts for several interior factors"""
import sys, os, time, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import bnb, sdpa_io, warm_bnb
name = sys.argv[1]
inst = sdpa_io.read_sdpa(os.path.join(ROOT, 'tests', 'golden', 'instances', name))
prob = bnb.instance_to_sdpi(inst)
objl = os.environ.get('BNB_OBJLIMIT') == '1'
for lam in [float(v) for v in sys.argv[2:]] or [0.0, 0.1, 0.3, 0.5]:
    s, solve, stats = warm_bnb.warm_node_solver(hb.lib(), 1e-6, lam, objl)
    t = time.time(); r = bnb.branch_and_bound(prob, inst.intvars, solve); t = time.time() - t
    s.free()
    print("%s lam %.2f: best %s nodes %d failed %d iterations %d (%.1f per node) warm starts %d cutoffs %d engine %.3f s wall %.2f s"
          % (name, lam, r[0], r[2], r[3], stats["iters"], stats["iters"] / max(1, stats["calls"]), stats["warm"], stats["cutoff"], stats["time"], t), flush=True)
