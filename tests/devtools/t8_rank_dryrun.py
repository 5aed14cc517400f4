#!/usr/bin/env python3
"""t8_rank_dryrun.py [n m ranks [sharded|replicated]] - developer tool: ONE rank of the variable-sharded solve at n=4000, m=8000 on 8 ranks, run alone on one
device through the measurement transport (collectives move nothing, so the numbers are meaningless): checks that everything this
rank allocates fits the device (its 128 GB of A, the packed copy, the slice buffers, the split-K slabs) and that set_shape, the
device-side generator and two iterations run at that size.  Prints the free device memory at each stage."""
import ctypes as C, importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
layout = sys.argv[4] if len(sys.argv) > 4 else "sharded"          # "replicated": column slices of the replicated matrices
lib = hb.lib()
def free_gb():
    f, t = torch.cuda.mem_get_info(0)
    return f / 1e9, t / 1e9
print("device memory: %.1f GB free of %.1f" % free_gb(), flush=True)
s = hb.Solver(0)
comm = C.c_void_p()
assert lib.hipsdp_comm_create_null(0, G, C.byref(comm)) == 0
assert lib.hipsdp_set_comm(s.h, comm, 0, G) == 0
assert lib.hipsdp_shard_matrices(s.h, 1 if layout == "sharded" else 0) == 0
t0 = time.perf_counter(); s.set_shape(m, [n], 0)
print("set_shape %.1f s, matrices sharded: %d, free %.1f GB" % (time.perf_counter() - t0, lib.hipsdp_matrices_sharded(s.h), free_gb()[0]), flush=True)
rng = np.random.default_rng(1)
t0 = time.perf_counter(); b = s.gen_planted(n, m, 20240, 0.5 * np.eye(n), np.eye(n), rng.uniform(-1, 1, m))
print("generator %.1f s, free %.1f GB" % (time.perf_counter() - t0, free_gb()[0]), flush=True)
t0 = time.perf_counter(); info = s.solve(gaptol=1e-5, feastol=1e-5, maxiter=int(os.environ.get("DRY_ITERS", "2")))
print("%d iterations in %.4f s (%.2f ms per iteration): status %d, %d assemblies in %.4f s (%.1f TFLOP/s algorithmic for this rank's 1/%d), free %.1f GB" % (
    info.iterations, info.solve_seconds, 1e3 * info.solve_seconds / max(1, info.iterations), info.status, info.schur_calls, info.schur_seconds,
    info.schur_flops / G / max(info.schur_seconds, 1e-9) / 1e12, G, free_gb()[0]), flush=True)
assert lib.hipsdp_set_comm(s.h, None, 0, 1) == 0
s.close(); lib.hipsdp_comm_destroy(comm)
