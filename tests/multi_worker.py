#!/usr/bin/env python3
"""multi_worker.py RANK WORLD SHM_NAME N M LPROWS OUT.json [LOAD [N2]] - one rank of a sharded node solve (test helper).

Every rank builds the same instance (as bench.py does at N > 1), joins the communicator and runs the engine with its share of
the Schur rows.  WORLD ranks may share one device through the host-staged communicator (RCCL refuses two ranks on one GPU).
The result every rank saw goes to OUT.json for tests/test_gpu_multi.py to compare.

LOAD = vars-dense | vars-coo | vars-gen: the constraint matrices are sharded by variable (hipsdp_shard_matrices) and loaded
through hipsdp_set_block_dense / hipsdp_add_entries / generated on the device by hipsdp_gen_planted (every rank passes the whole
instance; a rank stores the matrices it holds).  dense | coo | gen: the same loaders with replicated matrices.
N2 > 0 adds a second dense block of that size (dense and coo loaders)."""
import ctypes as C
import importlib.util
import json
import os
import sys

# the test problems are small: without this the engine would run them replicated (hipsdp.h, HIPSDP_SHARD_MIN_FLOPS) and the
# sharded code these tests are about would never execute
os.environ.setdefault("HIPSDP_SHARD_MIN_FLOPS", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "harness"))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hb)
import numpy as np
import ipm_ref
import instances


def main():
    rank, world, name, n, m, q, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), \
        int(sys.argv[6]), sys.argv[7]
    b, A, ys, Xs, Zs = instances.planted_dense(n, m)
    blocks = [A]
    n2 = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    if n2 > 0:
        # a second block with the same planted y: A0 = sum_i ys_i A_i - Zs2, b += A2(Xs2)
        b2, A2, _, Xs2, Zs2 = instances.planted_dense(n2, m, seed=4711)
        A2[0] = np.tensordot(ys, A2[1:], axes=(0, 0)) - Zs2
        A2[0] = 0.5 * (A2[0] + A2[0].T)
        b = b + A2[1:].reshape(m, -1) @ Xs2.reshape(-1)
        blocks.append(A2)
    if q > 0:
        rng = np.random.default_rng(5)
        D = rng.standard_normal((q, m))
        c = D @ ys - rng.uniform(0.1, 1.0, q)            # strictly satisfied at the planted y: the optimum does not move
        core = ipm_ref.CoreProblem(b, blocks, D, c)
    else:
        core = ipm_ref.CoreProblem(b, blocks)
    load = sys.argv[8] if len(sys.argv) > 8 else "dense"
    lib = hb.lib()
    s = hb.Solver(0)
    comm = C.c_void_p()
    if world > 1:
        staging = int(os.environ.get("HIPSDP_TEST_STAGING", str(1 << 20)))
        rc = lib.hipsdp_comm_create_host(name.encode(), rank, world, C.c_longlong(staging), C.c_double(60.0), C.byref(comm))
        assert rc == 0, rc
        assert lib.hipsdp_set_comm(s.h, comm, rank, world) == 0
        if load.startswith("vars-"):
            assert lib.hipsdp_shard_matrices(s.h, 1) == 0
    kind = load.split("-")[-1]
    if kind == "dense":
        s.load_core(core)
    elif kind == "coo":
        s.set_shape(core.m, [Ak.shape[1] for Ak in blocks], core.q)
        s.set_obj(core.b)
        for k, Ak in enumerate(blocks):
            nk = Ak.shape[1]
            v, r, c = np.nonzero(np.tril(np.ones((nk, nk)))[None, :, :] * (Ak != 0))
            s.add_entries(k, v.astype(np.int32), r.astype(np.int32), c.astype(np.int32), Ak[v, r, c])
        if core.q:
            s.set_lp(np.hstack([core.c[:, None], core.D]))
    else:
        assert q == 0 and n2 == 0
        s.set_shape(m, [n], 0)
        bg = s.gen_planted(n, m, 77, Xs, Zs, ys)
        b = bg
    kw = {}
    if os.environ.get("HIPSDP_TEST_TIMELIMIT"):
        kw["timelimit"] = float(os.environ["HIPSDP_TEST_TIMELIMIT"])
    info = s.solve(gaptol=1e-6, feastol=1e-6, **kw)
    y = s.y()
    X = s.X(0)
    if world > 1:
        assert lib.hipsdp_set_comm(s.h, None, 0, 1) == 0
    s.close()
    if world > 1:
        lib.hipsdp_comm_destroy(comm)
    json.dump(dict(rank=rank, status=int(info.status), iterations=int(info.iterations), pobj=float(info.pobj),
                   dobj=float(info.dobj), y=[float(v) for v in y], xtrace=float(np.trace(X)),
                   xfro=float(np.linalg.norm(X))), open(out, "w"))


if __name__ == "__main__":
    main()
