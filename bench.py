#!/usr/bin/env python3
"""bench.py - node-SDP solves/sec (and IPM iterations/sec) of the HIP interior-point engine on BASELINE.json's synthetic
dense block.  One "step" = one complete node solve (cold start -> optimal to gaptol = feastol = 1e-5) of the planted
instance of BASELINE.md section 3, with A already resident in HBM (generated on the device).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 500] [--m 1000] [--no-cpu]

N > 1 is launched by torch.distributed.run (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE from the environment).
Rank 0 prints ONE JSON line.  The oracle (oracle/ipm_ref.py) is imported only inside cpu_baseline().
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (vendor figure; rocBLAS dgemm reaches 72.8 on this pool)
# HBM bytes per Schur assembly from the PMC passes committed under profiles/ (FETCH_SIZE doubled as MI355X_MICROARCH.md
# prescribes for gfx950, plus WRITE_SIZE), keyed by (n, m); None when not measured for a size
TRAFFIC_BYTES_PER_ASSEMBLY = {(500, 1000): 18.79e9}      # profiles/r01_h_pmc_traffic_c2.txt


def load_binding():
    spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def counter_uniform(seed, idx):
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + idx.astype(np.uint64))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def counter_normal(seed, idx):
    u1 = counter_uniform(seed, np.uint64(2) * idx.astype(np.uint64))
    u2 = counter_uniform(seed, np.uint64(2) * idx.astype(np.uint64) + np.uint64(1))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def planted_pair(n, m, seed):
    """complementary (X*, Z*) with rank n/4 and n - n/4, eigenvalues U[1,2]; y* ~ U[-1,1]"""
    Q, _ = np.linalg.qr(counter_normal(seed + 1000003, np.arange(n * n, dtype=np.uint64)).reshape(n, n))
    r = max(1, int(round(n * 0.25)))
    ev = 1.0 + counter_uniform(seed + 2000003, np.arange(n, dtype=np.uint64))
    Xs = (Q * np.where(np.arange(n) < r, ev, 0.0)) @ Q.T
    Zs = (Q * np.where(np.arange(n) < r, 0.0, ev)) @ Q.T
    ys = 2.0 * counter_uniform(seed + 3000003, np.arange(m, dtype=np.uint64)) - 1.0
    return Xs, Zs, ys


def cpu_baseline(solver, b, n, m, gpu_iterations, budget_iters=None):
    """CPU restatement (numpy + OpenBLAS threads) of the same algorithm on the same bits: a bounded sample of IPM iterations
    timed on the host cores, extrapolated to the iteration count of the full solve."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ipm_ref
    if budget_iters is None:
        # about 10-30 s of CPU work: 4 iterations at C2 (7.5e11 flop each), 1 at T1 (1.2e13)
        budget_iters = 4 if 4.0 * m * n ** 3 + float(m) ** 2 * n ** 2 < 2e12 else 1
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    A = solver.get_block_dense(0)
    core = ipm_ref.CoreProblem(b, [A])
    t0 = time.perf_counter()
    res = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=1e-5, feastol=1e-5, maxiter=budget_iters))
    dt = time.perf_counter() - t0
    its = max(1, res.iterations)
    per_iter = dt / its
    solves_per_sec = 1.0 / (per_iter * max(1, gpu_iterations))
    return {"value": solves_per_sec, "unit": "solves/s", "cores": int(threads), "kind": "port",
            "iters_per_sec": 1.0 / per_iter,
            "sample": "%d IPM iterations of the same n=%d, m=%d instance (A copied back from HBM: identical bits) with "
                      "oracle/ipm_ref.py on numpy/OpenBLAS, %.1f s; solves/s extrapolated to the %d iterations of the full solve"
                      % (its, n, m, dt, gpu_iterations)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--m", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=20240)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--shard-matrices", choices=["auto", "on", "off"], default="auto",
                    help="N > 1: constraint matrices sharded by variable (auto: when the replicated matrices would not fit)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl")
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)

    hb = load_binding()
    if hb.device_count() <= 0:
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible and hipsdp has no CPU path")
    n, m = args.n, args.m
    solver = hb.Solver(local_rank if world > 1 else 0)
    # N > 1: ONE node SDP, its Schur assembly sharded over the ranks (north_star); every rank holds the same instance - or,
    # when the replicated constraint matrices would not fit the device (n=4000, m=8000: 1 TB), only the matrices of its variables
    comm = None
    if world > 1:
        import ctypes as C
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            assert hb.lib().hipsdp_comm_unique_id(buf) == 0
            uid = torch.tensor(list(buf), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, src=0)
        raw = (C.c_ubyte * 128)(*uid.cpu().tolist())
        comm = C.c_void_p()
        rc = hb.lib().hipsdp_comm_create(raw, rank, world, C.byref(comm))
        assert rc == 0, "hipsdp_comm_create failed: %s" % hb.lib().hipsdp_last_error().decode()
        assert hb.lib().hipsdp_set_comm(solver.h, comm, rank, world) == 0
        assert hb.lib().hipsdp_shard_matrices(solver.h, {"auto": -1, "on": 1, "off": 0}[args.shard_matrices]) == 0
    solver.set_shape(m, [n], 0)
    Xs, Zs, ys = planted_pair(n, m, args.seed)
    b = solver.gen_planted(n, m, args.seed, Xs, Zs, ys)
    opt = float(b @ ys)

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    infos = []
    for _ in range(args.warmup):
        solver.solve(gaptol=1e-5, feastol=1e-5)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        infos.append(solver.solve(gaptol=1e-5, feastol=1e-5))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    last = infos[-1]
    ok = all(i.status == 0 for i in infos) and abs(last.dobj - opt) <= 1e-5 * (1 + abs(opt))
    iters = sum(i.iterations for i in infos)
    schur_s = sum(i.schur_seconds for i in infos)
    schur_fl = sum(i.schur_flops for i in infos)
    schur_calls = sum(i.schur_calls for i in infos)
    achieved = schur_fl / max(schur_s, 1e-12) / 1e12
    out = {
        "metric": "node-SDP solves/sec, dense block n x n with m vars (IPM iters/sec in iters_per_sec)",
        "value": args.steps / elapsed,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: synthetic single dense block n=%d, m=%d, fp64, planted optimum, cold start, "
                               "gaptol=feastol=1e-5, A resident in HBM; N > 1: Schur assembly (column slices of W_j = G A_j R, RCCL all-reduce of the partial Schur matrices) and the passes over A (by rows, RCCL all-gather / all-reduce) sharded over the GPUs, everything else replicated" % (n, m),
                   "parallelism": "schur-shards x%d" % world,
                   "matrices": "sharded by variable (W_j formed where A_j lives, all-to-all of the W entries, all-reduce of the partial "
                               "Schur matrices)" if hb.lib().hipsdp_matrices_sharded(solver.h) else "replicated",
                   "n": n, "m": m, "seed": args.seed},
        "iters_per_sec": iters / elapsed,
        "iterations_per_solve": iters / max(1, len(infos)),
        "solution_check": {"status_optimal_and_objective_matches_planted_optimum": bool(ok), "objective": last.dobj,
                           "planted_optimum": opt, "pinf": last.pinf, "dabs": last.dabs, "gap": last.gap},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                     "frac": achieved / (FP64_MFMA_PEAK_TFLOPS * world), "traffic": TRAFFIC_BYTES_PER_ASSEMBLY.get((n, m)),
                     "kernel": "hs_dgemm2_kernel (Schur assembly: stack GEMM, batched GEMM, K-sliced Gram GEMM + slice reduce)",
                     "algorithmic_flops_per_assembly": schur_fl / max(1, schur_calls),
                     "avg_assembly_ms": 1e3 * schur_s / max(1, schur_calls),
                     "assemblies": schur_calls,
                     "schur_share_of_solve_time": schur_s / max(1e-12, sum(i.solve_seconds for i in infos))},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(solver, b, n, m, int(round(iters / max(1, len(infos)))))
    elif rank == 0:
        out["cpu_baseline"] = None
    if comm is not None:
        hb.lib().hipsdp_set_comm(solver.h, None, 0, 1)
    solver.close()
    if comm is not None:
        hb.lib().hipsdp_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if not ok:
        sys.exit(2)


if __name__ == "__main__":
    main()
