#!/usr/bin/env python3
"""bench.py - node-SDP solves/sec (and IPM iterations/sec) of the HIP interior-point engine on BASELINE.json's synthetic
dense block.  One "step" = one complete node solve (cold start -> optimal to gaptol = feastol = 1e-5) of the planted
instance of BASELINE.md section 3, with A already resident in HBM (generated on the device).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 500] [--m 1000] [--no-cpu]

N > 1: one process per GPU.  When the driver has already started the ranks (torch.distributed.run: RANK / LOCAL_RANK /
WORLD_SIZE in the environment) this process is one of them and checks that --gpus equals WORLD_SIZE.  Started plainly as
`python bench.py --gpus N`, it is the PARENT: before anything touches the GPU it starts the N ranks as child processes
(python -m torch.distributed.run ... bench.py <same arguments>), waits for them, relays rank 0's JSON line and exits
non-zero when any rank failed.  Rank 0 prints ONE JSON line.  The oracle (oracle/ipm_ref.py) is imported only inside
cpu_baseline().
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (vendor figure; rocBLAS dgemm reaches 72.8 on this pool)
# Numbers that only the PMC passes of rocprofv3 can give (a live bench run cannot collect counters): HBM bytes per Schur
# assembly (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE), MFMA busy share and EXECUTED
# FP64 matrix rate of the two n^3 products / the Gram product.  Keyed by (n, m), quoted from the committed file named in
# "source" and labelled as such in the JSON line; one GPU only; None for sizes that were not profiled.
PMC_FROM_PROFILES = {
    (500, 1000): {"traffic_bytes_per_assembly": 23.32e9, "mfma_busy": {"n3_products": 0.755, "gram": 0.842},
                  "executed_tflops": {"n3_products": 49.8, "gram": 54.8},
                  "traffic_is": "L2-miss traffic (FETCH_SIZE x 2 + WRITE_SIZE) of one GENERAL assembly: 6.69 + 7.17 GB for the two n^3 products "
                                "(8.21 + 7.47 before their list was walked in sets of panels: profiles/r05_gemm_order_traffic.txt), "
                                "9.46 GB for the Gram product (whose own kernel misses on 37 % of its L2 requests, the tile kernel it replaced on 14 %, and is "
                                "9 % faster: profiles/r06_pmc_gram_l2.txt, profiles/r05_gram_l2_miss_experiments.txt); algorithmic 10 GB",
                  "source": "profiles/r06_a_pmc_traffic_c2.txt, profiles/r06_a_pmc_mfma_util.txt (counter passes run the kernels 2-8 % slower and at "
                            "lower clocks than the timed solve; same-box A/B of the two list orders: profiles/r05_gemm_order_traffic.txt)"},
    (1000, 2000): {"traffic_bytes_per_assembly": None, "mfma_busy": {"n3_products": 0.822, "gram": 0.872},
                   "executed_tflops": {"n3_products": 59.9, "gram": 67.6},
                   "source": "profiles/r06_a_pmc_mfma_util.txt"},
}
WORKLOAD_NAMES = {(500, 1000): "BASELINE configs[1] (C2)", (1000, 2000): "BASELINE.md T1 (north_star target size)",
                  (2000, 4000): "BASELINE configs[3] (C4)", (4000, 8000): "BASELINE.md T8"}


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


def host_cpu_share():
    """cores this process may actually use: the affinity mask, cut by the cgroup's CPU quota (a GPU box hands a job a share of its
    host cores; BLAS threads beyond it only fight each other)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(solver, b, n, m, gpu_iterations, budget_iters=None):
    """The same algorithm on the same bits on the host cores: oracle/cpu_ref_dense.c - the C restatement of oracle/ipm_ref.py for one dense
    block, Schur assembly in the device path's W formulation (W_j = G A_j R by two DTRMM, Mx = W W^T as DGEMM panels), OpenMP over
    independent pieces, scipy's bundled OpenBLAS single-threaded underneath (kind "own C restatement + OpenBLAS").  budget_iters = None:
    the WHOLE solve is timed (the bench size: about 10 s); a number: that many iterations are timed and the solves/s are extrapolated to
    the iteration count of the device's solve (n = 1000 / m = 2000: 8e12 flop per iteration).  Without an OpenBLAS in the image: the numpy
    port of rounds 1-5 (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    threads = host_cpu_share()
    A = solver.get_block_dense(0)
    m1 = m + 1
    try:
        import cpu_ref_dense
        cpu_ref_dense.build()
    except Exception:
        cpu_ref_dense = None
    if cpu_ref_dense is not None:
        full = budget_iters is None
        info, _y = cpu_ref_dense.solve(b, A, gaptol=1e-5, feastol=1e-5, maxiter=(200 if full else budget_iters), threads=threads)
        its = max(1, info.iterations)
        per_iter = info.total_seconds / its
        if full:
            solves_per_sec = 1.0 / info.total_seconds
            sample = ("one complete solve of the same n=%d, m=%d instance (A copied back from HBM: identical bits): %d iterations, status %d, "
                      "objective %.9g, %.1f s of which %.1f s Schur assembly; nothing extrapolated" %
                      (n, m, info.iterations, info.status, info.dobj, info.total_seconds, info.schur_seconds))
        else:
            solves_per_sec = 1.0 / (per_iter * max(1, gpu_iterations))
            sample = ("%d IPM iterations of the same n=%d, m=%d instance (A copied back from HBM: identical bits), %.1f s of which %.1f s "
                      "Schur assembly; solves/s extrapolated to the %d iterations of the device's solve" %
                      (its, n, m, info.total_seconds, info.schur_seconds, gpu_iterations))
        return {"value": solves_per_sec, "unit": "solves/s", "cores": int(threads), "kind": "own C restatement + OpenBLAS",
                "blas": cpu_ref_dense.blas_name() + " (single-threaded calls under OpenMP, %d threads)" % threads,
                "source": "oracle/cpu_ref_dense.c", "iterations": int(info.iterations), "extrapolated": (not full),
                "iters_per_sec": 1.0 / per_iter,
                "executed_tflops_of_the_assembly_formulation":
                    (2.0 * m1 * n ** 3 + float(m1) ** 2 * n ** 2) * its / max(info.schur_seconds, 1e-9) / 1e12,
                "sample": sample}
    import ipm_ref
    if budget_iters is None:
        budget_iters = 4 if 4.0 * m * n ** 3 + float(m) ** 2 * n ** 2 < 2e12 else 1
    core = ipm_ref.CoreProblem(b, [A])
    par = ipm_ref.Params(gaptol=1e-5, feastol=1e-5, maxiter=budget_iters)
    par.schur = "W"
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    if threadpool_limits is not None:
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            res = ipm_ref.hsd_solve(core, par)
            dt = time.perf_counter() - t0
    else:
        threads = os.cpu_count() or threads
        t0 = time.perf_counter()
        res = ipm_ref.hsd_solve(core, par)
        dt = time.perf_counter() - t0
    its = max(1, res.iterations)
    per_iter = dt / its
    solves_per_sec = 1.0 / (per_iter * max(1, gpu_iterations))
    return {"value": solves_per_sec, "unit": "solves/s", "cores": int(threads), "kind": "port",
            "iters_per_sec": 1.0 / per_iter, "extrapolated": True,
            "executed_tflops_of_the_assembly_formulation": (2.0 * m1 * n ** 3 + float(m1) ** 2 * n ** 2) / per_iter / 1e12,
            "sample": "%d IPM iterations of the same n=%d, m=%d instance with oracle/ipm_ref.py on numpy (no OpenBLAS to link the C "
                      "restatement against), %.1f s; solves/s extrapolated to the %d iterations of the full solve" % (its, n, m, dt, gpu_iterations)}


def workload_name(n, m):
    return WORKLOAD_NAMES.get((n, m), "custom size (not a BASELINE configuration)")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--m", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=20240)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--density", type=float, default=0.1,
                    help="density of the matrices in the `density` sub-object (SURVEY.md 8(d): 0.1); the headline run is dense")
    ap.add_argument("--no-extras", action="store_true", help="skip the t1 / sdpi_boundary / warm_start / phases sub-objects")
    ap.add_argument("--shard-matrices", choices=["auto", "on", "off"], default="auto",
                    help="N > 1: constraint matrices sharded by variable (auto: when the replicated matrices would not fit)")
    ap.add_argument("--schur-form", choices=["auto", "cols", "rows", "vars"], default="auto",
                    help="N > 1, how the Schur assembly is sharded: cols = column slices of W_j = G A_j R + all-reduce (default when the "
                         "matrices fit replicated), rows = row chunks of the Schur matrix + RCCL all-gather (the form BASELINE.json's "
                         "north_star words), vars = matrices sharded by variable + all-to-all of the W entries (default when they do not fit)")
    ap.add_argument("--master-port", type=int, default=0, help="parent mode: rendezvous port of the ranks (0: pick a free one)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="exercise the N-rank launcher without a GPU: the ranks meet over gloo and rank 0 prints a line")
    return ap.parse_args(argv)


# ---- parent mode: `python bench.py --gpus N` without a launcher ---------------------------------------------------------

def visible_gpus():
    """Number of GPUs WITHOUT loading the HIP runtime into this process (it is the parent of the ranks: a process that has
    initialised the GPU must not start the launcher).  The KFD topology in sysfs lists one node per agent; GPU nodes have
    simd_count > 0.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restrict what a rank will see.  Where sysfs has no KFD tree the
    count comes from a throw-away child process."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    for f in nodes:
        try:
            for ln in open(f):
                k, _, v = ln.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
                    break
        except (OSError, ValueError):
            pass
    if not nodes:
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=300)
            n = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
        except Exception:
            n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def hip_runtime_mapped():
    """whether libamdhip64 is mapped into this process (the parent of the ranks must answer False: tests/test_bench_launcher_cpu.py)"""
    try:
        return any("libamdhip64" in ln for ln in open("/proc/self/maps"))
    except OSError:
        return False


def launch_ranks(args, argv):
    """Starts the N ranks as CHILD processes (never exec: nothing in this process may have touched the GPU, and it has not),
    relays rank 0's JSON line, returns the exit code."""
    import socket
    import subprocess
    n = args.gpus
    if not args.launch_dry_run:
        have = visible_gpus()
        if os.environ.get("BENCH_REPORT_MAPS"):            # test hook: what the parent has mapped at the point where it would start the ranks
            sys.stderr.write("bench.py: parent maps: %s\n" % json.dumps({"hip_mapped": hip_runtime_mapped(), "torch_imported": "torch" in sys.modules,
                                                                         "visible_gpus": have}))
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, but only %d GPU(s) are visible on this machine: a %d-rank RCCL run "
                             "needs %d devices (one process per GPU); nothing was run\n" % (n, have, n, n))
            return 3
    port = args.master_port
    if port <= 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    out, _ = proc.communicate()
    line = None
    for ln in out.splitlines():
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                json.loads(t)
                line = t
            except ValueError:
                pass
        elif t:
            sys.stderr.write(ln + "\n")
    if proc.returncode != 0:
        sys.stderr.write("bench.py: the %d-rank run failed (launcher exit code %d)\n" % (n, proc.returncode))
        if line is not None:
            sys.stderr.write("bench.py: partial result of rank 0: %s\n" % line)
        return proc.returncode if proc.returncode > 0 else 1
    if line is None:
        sys.stderr.write("bench.py: the ranks finished but rank 0 printed no JSON line\n")
        return 4
    print(line)
    return 0


def dry_run_rank(args):
    """--launch-dry-run: what a rank does up to the first collective, without a GPU (CPU test of the launcher)"""
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    if os.environ.get("BENCH_DRYRUN_FAIL_RANK") == str(rank):      # test hook: a rank that dies must fail the whole run
        dist.destroy_process_group()
        return 7
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "gpus_arg": args.gpus, "rank_sum": float(t.item()), "schur_form": args.schur_form,
                          "local_rank": int(os.environ.get("LOCAL_RANK", "-1")), "master_addr": os.environ.get("MASTER_ADDR")}))
    dist.destroy_process_group()
    return 0


# ---- sub-benchmarks of the one JSON line ---------------------------------------------------------------------------------

def run_solves(solver, steps, warmup, barrier, **kw):
    for _ in range(warmup):
        solver.solve(gaptol=1e-5, feastol=1e-5, **kw)
    barrier()
    t0 = time.perf_counter()
    infos = [solver.solve(gaptol=1e-5, feastol=1e-5, **kw) for _ in range(steps)]
    barrier()
    return infos, time.perf_counter() - t0


def schur_summary(infos, n, m, world=1):
    """roofline of the dominant kernels (the Schur assembly).  frac = EXECUTED matrix-core flops / time / peak: what the engine's GEMM
    launches issue (whole tiles over the K ranges actually walked - hipsdp_info.schur_flops_executed), i.e. the number MFMA
    utilisation counters measure; the algorithmic count of SURVEY.md 8(d) (4 m1 n^3 + m1^2 n^2, which the W formulation undercuts)
    is reported beside it."""
    schur_s = sum(i.schur_seconds for i in infos)
    schur_alg = sum(i.schur_flops for i in infos)
    schur_exe = sum(i.schur_flops_executed for i in infos)
    calls = sum(i.schur_calls for i in infos)
    peak = FP64_MFMA_PEAK_TFLOPS * world
    ach = schur_exe / max(schur_s, 1e-12) / 1e12
    alg = schur_alg / max(schur_s, 1e-12) / 1e12
    return {"achieved": ach, "frac": ach / peak, "avg_assembly_ms": 1e3 * schur_s / max(1, calls),
            "assemblies": calls, "executed_flops_per_assembly": schur_exe / max(1, calls),
            "algorithmic_flops_per_assembly": schur_alg / max(1, calls),
            "algorithmic_equivalent_tflops": alg,
            "schur_share_of_solve_time": schur_s / max(1e-12, sum(i.solve_seconds for i in infos))}


def phase_anatomy(hb, solver):
    """one extra (untimed) solve with HIP events at the phase boundaries of the engine's stream (also roctx ranges)"""
    import ctypes as C
    lib = hb.lib()
    lib.hipsdp_phase_name.restype = C.c_char_p
    lib.hipsdp_set_profiling(solver.h, 1)
    info = solver.solve(gaptol=1e-5, feastol=1e-5)
    ms = (C.c_double * 8)()
    lib.hipsdp_get_phase_times(solver.h, ms)
    lib.hipsdp_set_profiling(solver.h, 0)
    its = max(1, info.iterations)
    out = {lib.hipsdp_phase_name(p).decode(): ms[p] / its for p in range(7)}
    out["unit"] = "ms per iteration (device time between HIP events on the engine's stream)"
    out["non_schur_ms_per_iteration"] = sum(ms[p] for p in range(7) if p != 2) / its
    return out


def bench_t1(hb, seed, barrier, cpu=False):
    """n = 1000, m = 2000: the size north_star states its 1-GPU target on (>= 10x CPU, >= 30 % FP64-MFMA in the assembly)"""
    n, m = 1000, 2000
    s = hb.Solver(0)
    try:
        s.set_shape(m, [n], 0)
        Xs, Zs, ys = planted_pair(n, m, seed)
        b = s.gen_planted(n, m, seed, Xs, Zs, ys)
        opt = float(b @ ys)
        infos, el = run_solves(s, 2, 1, barrier)
        ok = all(i.status == 0 for i in infos) and abs(infos[-1].dobj - opt) <= 1e-5 * (1 + abs(opt))
        its = sum(i.iterations for i in infos)
        out = {"workload": workload_name(n, m), "n": n, "m": m, "steps": len(infos), "solves_per_sec": len(infos) / el,
               "ms_per_step": 1e3 * el / len(infos), "iters_per_sec": its / el, "iterations_per_solve": its / len(infos),
               "matches_planted_optimum": bool(ok), "roofline": schur_summary(infos, n, m)}
        pmc = PMC_FROM_PROFILES.get((n, m))
        if pmc is not None:
            out["roofline"]["from_committed_pmc_profile"] = pmc
        if cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(s, b, n, m, int(round(its / len(infos))), budget_iters=2)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        return out
    finally:
        s.close()


def bench_density(hb, n, m, seed, density, barrier):
    """SURVEY.md 8(d)'s rho = 0.1 variant of the same workload: the matrices A_i have that density (device generator), everything else
    as in the headline run.  At this density the dense formulation is still the cheaper one (4 (sum nnz)^2 = 6e16 against 7.5e11), so
    the block stays dense and the time per solve is the headline's; the sparse storage of csrc/sparse.hip takes over where the
    reference's real instances live (a few nonzeros per matrix): second entry, n = 500, m = 2000, 3 nonzeros per matrix."""
    out = {}
    s = hb.Solver(0)
    try:
        s.set_shape(m, [n], 0)
        Xs, Zs, ys = planted_pair(n, m, seed)
        b = s.gen_planted_density(n, m, seed, density, Xs, Zs, ys)
        opt = float(b @ ys)
        infos, el = run_solves(s, 3, 1, barrier)
        out["dense_block_density_%g" % density] = {
            "n": n, "m": m, "solves_per_sec": len(infos) / el, "iterations_per_solve": sum(i.iterations for i in infos) / len(infos),
            "matches_planted_optimum": bool(all(i.status == 0 for i in infos) and abs(infos[-1].dobj - opt) <= 1e-5 * (1 + abs(opt))),
            "storage": "dense (cost rule: pair formula %.1e multiply-adds against %.1e)" % (
                4.0 * (density * n * (n + 1) / 2 * m) ** 2, 4.0 * (m + 1) * n ** 3 + float(m + 1) ** 2 * n ** 2)}
    finally:
        s.close()
    # the sparse regime: triplets from the host (6000 of them), kept as nonzeros by the engine
    rng = np.random.default_rng(seed)
    n2, m2, k = 500, 2000, 3
    var = np.repeat(np.arange(1, m2 + 1, dtype=np.int32), k)
    r = rng.integers(0, n2, size=m2 * k)
    c = rng.integers(0, n2, size=m2 * k)
    c[::k] = r[::k]                                                  # one diagonal entry per matrix
    row, col = np.maximum(r, c).astype(np.int32), np.minimum(r, c).astype(np.int32)
    key = var.astype(np.int64) * n2 * n2 + row.astype(np.int64) * n2 + col
    _, first = np.unique(key, return_index=True)
    var, row, col = var[first], row[first], col[first]
    val = rng.standard_normal(len(var))
    Xs, Zs, ys = planted_pair(n2, m2, seed + 1)
    A0 = -Zs.copy()
    w = val * ys[var - 1]
    np.add.at(A0, (row, col), w)
    off = row != col
    np.add.at(A0, (col[off], row[off]), w[off])
    A0 = 0.5 * (A0 + A0.T)
    b = np.bincount(var - 1, weights=val * Xs[row, col] * np.where(off, 2.0, 1.0), minlength=m2)
    opt = float(b @ ys)
    s = hb.Solver(0)
    try:
        s.load_sparse(m2, n2, b, (var, row, col, val), A0)
        infos, el = run_solves(s, 5, 1, barrier)
        out["sparse_block_3_nonzeros_per_matrix"] = {
            "n": n2, "m": m2, "kept_as_nonzeros": bool(s.is_sparse(0)), "solves_per_sec": len(infos) / el,
            "iterations_per_solve": sum(i.iterations for i in infos) / len(infos),
            "matches_planted_optimum": bool(all(i.status == 0 for i in infos) and abs(infos[-1].dobj - opt) <= 1e-5 * (1 + abs(opt))),
            "dense_storage_would_be_GB": 8.0 * (m2 + 1) * n2 * n2 / 1e9}
    finally:
        s.close()
    return out


def bench_warm_start(solver, n, m, b, barrier, cold_iters):
    """warm-start variant of SURVEY.md section 8(d): the start point is the optimum of the node pushed into the interior the way
    relax_sdp's warmstartipfactor rule does it (convex combination with a scaled identity, factor 0.5) and y perturbed by 1e-3"""
    y = solver.y()
    X = solver.X(0)
    Z = solver.Z(0)
    rng = np.random.default_rng(7)
    y0 = y + 1e-3 * rng.standard_normal(m)
    lam = 0.5
    sx = max(1.0, float(np.trace(X)) / n)
    sz = max(1.0, float(np.trace(Z)) / n)
    X0 = (1 - lam) * X + lam * sx * np.eye(n)
    Z0 = (1 - lam) * Z + lam * sz * np.eye(n)
    times, its, used = [], [], []
    for _ in range(3):
        solver.set_start(y0, [X0], [Z0])
        barrier()
        t0 = time.perf_counter()
        info = solver.solve(gaptol=1e-5, feastol=1e-5)
        barrier()
        times.append(time.perf_counter() - t0)
        its.append(info.iterations)
        used.append(int(info.warm_started))
    opt_ok = info.status == 0
    return {"start_point": "optimum of the same node, X and Z moved half-way to a scaled identity, y perturbed by 1e-3 (parent-node "
                           "stand-in); the upload of the start point (2 x n^2 doubles) is outside the timed region",
            "solves_per_sec": 1.0 / (sum(times[1:]) / len(times[1:])), "iterations_per_solve": its[-1],
            "cold_start_iterations": cold_iters, "start_point_used_by_engine": bool(all(used)), "status_optimal": bool(opt_ok)}


def bench_sdpi_boundary(hb, solver, n, m, b, opt):
    """The same node through the drop-in boundary SCIPsdpiSolverLoadAndSolve (48-argument call of sdpisolver.h:176-233): the
    caller's COO arrays (lower triangles, 1.25e8 triplets at C2 = 2 GB) are host buffers, so the first call pays the PCIe upload
    into the device-resident master copy; later calls (the other nodes of a tree) find it by fingerprint."""
    import ctypes as C
    lib = hb.lib()
    A = solver.get_block_dense(0)                  # identical bits to the engine-level bench
    il = np.tril_indices(n)
    rows = np.ascontiguousarray(il[0], dtype=np.int32)
    cols = np.ascontiguousarray(il[1], dtype=np.int32)
    nnz_per = len(rows)
    vals = np.ascontiguousarray(A[1:, il[0], il[1]])                       # [m, nnz_per]
    cval = np.ascontiguousarray(A[0, il[0], il[1]])
    keepc = np.abs(cval) > 0.0
    crow_a, ccol_a, cval_a = rows[keepc].copy(), cols[keepc].copy(), cval[keepc].copy()
    del A
    PD, PI = C.POINTER(C.c_double), C.POINTER(C.c_int)
    PPD, PPI = C.POINTER(PD), C.POINTER(PI)
    pi = lambda a: a.ctypes.data_as(PI)
    pd = lambda a: a.ctypes.data_as(PD)
    prow = (PI * m)(*[pi(rows)] * m)
    pcol = (PI * m)(*[pi(cols)] * m)
    pval = (PD * m)(*[vals[j].ctypes.data_as(PD) for j in range(m)])
    srow, scol, sval = (PPI * 1)(prow), (PPI * 1)(pcol), (PPD * 1)(pval)
    nvarnonz = np.full(m, nnz_per, dtype=np.int32)
    sdpvar = np.arange(m, dtype=np.int32)
    pnn, pvar = (PI * 1)(pi(nvarnonz)), (PI * 1)(pi(sdpvar))
    obj = np.ascontiguousarray(b, dtype=np.float64)
    lb = np.full(m, -1e20)
    ub = np.full(m, 1e20)
    sizes = np.array([n], dtype=np.int32)
    nbv = np.array([m], dtype=np.int32)
    constn = np.array([len(cval_a)], dtype=np.int32)
    pcr, pcc, pcv = (PI * 1)(pi(crow_a)), (PI * 1)(pi(ccol_a)), (PD * 1)(pd(cval_a))
    indch = np.zeros(n, dtype=np.int32)
    pind = (PI * 1)(pi(indch))
    nrem = np.zeros(1, dtype=np.int32)
    bic = np.zeros(1, dtype=np.int32)
    dummy_i = np.zeros(1, dtype=np.int32)
    dummy_d = np.zeros(1)
    h = C.c_void_p()
    assert lib.SCIPsdpiSolverCreate(C.byref(h), None, None, None) == 1
    for par, val in ((1, 1e-5), (2, 1e-5), (3, 1e-5)):        # GAPTOL, FEASTOL, SDPSOLVERFEASTOL as relax_sdp.c:70-71 sets them
        lib.SCIPsdpiSolverSetRealpar(h, par, C.c_double(val))

    def call():
        t0 = time.perf_counter()
        rc = lib.SCIPsdpiSolverLoadAndSolve(
            h, C.c_int(m), pd(obj), pd(lb), pd(ub), C.c_int(1), pi(sizes), pi(nbv),
            C.c_int(int(constn[0])), pi(constn), pcr, pcc, pcv,
            C.c_int(int(m * nnz_per)), pnn, pvar, srow, scol, sval,
            pind, pi(nrem), pi(bic), C.c_int(0),
            C.c_int(0), pi(dummy_i), pd(dummy_d), pd(dummy_d), C.c_int(0), pi(dummy_i), pi(dummy_i), pd(dummy_d),
            None, None, None, None, None, None, None, None, None,
            C.c_int(-1), C.c_double(1e20), None)
        dt = time.perf_counter() - t0
        o = C.c_double(0.0)
        ok = rc == 1 and bool(lib.SCIPsdpiSolverIsOptimal(h)) and lib.SCIPsdpiSolverGetObjval(h, C.byref(o)) == 1
        it, calls = C.c_int(0), C.c_int(0)
        lib.SCIPsdpiSolverGetIterations(h, C.byref(it))
        lib.SCIPsdpiSolverGetSdpCalls(h, C.byref(calls))
        return dt, ok, o.value, it.value, calls.value

    try:
        first = call()
        later = [call() for _ in range(3)]
    finally:
        lib.SCIPsdpiSolverFree(C.byref(h))
    tl = sum(x[0] for x in later) / len(later)
    good = all(x[1] and abs(x[2] - opt) <= 1e-5 * (1 + abs(opt)) for x in [first] + later)
    return {"entry_point": "SCIPsdpiSolverLoadAndSolve (sdpisolver.h:176-233), 1 dense block, %d COO triplets from host arrays" % (m * nnz_per),
            "first_call_s": first[0], "first_call_includes": "fingerprint + 2.25 GB upload of the master copy (caller's per-variable arrays "
            "streamed through pinned chunks) + device gather + solve + the backend's check of y (Cholesky certificate) - "
            "PCIe-inclusive, never `value`",
            "cached_call_s": tl, "cached_solves_per_sec": 1.0 / tl, "cached_call_includes": "fingerprint of the caller's arrays, device "
            "gather of the node's block, solve, check of y", "iterations": later[-1][3], "sdpcalls": later[-1][4],
            "optimal_and_matches_planted_optimum": bool(good)}


def bench_bnb(hb, cpu=True):
    """BASELINE configs[2] / configs[4]: full branch-and-bound over example_TT (truss topology, n=10, m=37, 85 LP rows) and example_CLS
    (n=43, m=33) with every node relaxation solved by the device engine through SCIPsdpiSolverLoadAndSolve, cold starts and warm starts
    (child from its parent's (y, Z, X), relax_sdp.c's warmstartipfactor rule with factor 0.5).  The tree search, the .dat-s reader and
    the sdpi.c-style argument preparation are the test harness (tests/harness: host-side driver code, no arithmetic of the path); node
    solves per second = node relaxations / wall time inside SCIPsdpiSolverLoadAndSolve.  CPU figure beside it: the same trees with the
    numpy restatement of the engine (oracle/ipm_ref.py) as node solver."""
    for d in ("tests", os.path.join("tests", "harness")):
        if os.path.join(ROOT, d) not in sys.path:
            sys.path.insert(0, os.path.join(ROOT, d))
    import bnb
    import sdpa_io
    import warm_bnb
    inst_dir = os.path.join(ROOT, "tests", "golden", "instances")
    solu = {"example_TT.dat-s.gz": 2.11803, "example_CLS.dat-s.gz": 7.1485}        # check/testset/short.solu
    out = {"unit": "node SDP solves/s (wall time inside SCIPsdpiSolverLoadAndSolve), tolerance 1e-6, 1 GPU"}
    for name in sorted(solu):
        inst = sdpa_io.read_sdpa(os.path.join(inst_dir, name))
        prob = bnb.instance_to_sdpi(inst)
        row = {}
        for label, lam in (("cold", 0.0), ("warm", 0.5)):
            s, solve, stats = warm_bnb.warm_node_solver(hb.lib(), 1e-6, lam)
            hb.lib().hipsdp_solve1_solves.restype = C.c_longlong
            one0 = hb.lib().hipsdp_solve1_solves()
            t0 = time.perf_counter()
            best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, solve)
            wall = time.perf_counter() - t0
            s.free()
            calls = max(1, stats["calls"])
            row[label] = {"optimum": best, "matches_short_solu": bool(best is not None and abs(best - solu[name]) <= 1e-4 * max(1.0, abs(solu[name]))),
                          "nodes": nodes, "node_solves": stats["calls"], "unresolved_nodes": failed,
                          "ipm_iterations_per_node": stats["iters"] / calls, "warm_started_nodes": stats["warm"],
                          "node_solves_per_sec": calls / max(stats["wall"], 1e-9), "ms_per_ipm_iteration": 1e3 * stats["time"] / max(1, stats["iters"]),
                          "engine_seconds": stats["time"], "load_and_solve_seconds": stats["wall"], "tree_wall_seconds": wall,
                          # engine solves (re-solves of the tolerance loop and ladder rungs included) that ran as ONE launch of one
                          # workgroup (csrc/solve1.hip); the rest took the general path (dense matrices: example_CLS)
                          "one_launch_engine_solves": int(hb.lib().hipsdp_solve1_solves() - one0)}
        if cpu:
            try:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import ipm_ref                                        # noqa: F401  (the CPU leg: numpy node solver of the harness)
                solve = bnb.oracle_node_solver(1e-6)
                cnt = {"calls": 0, "t": 0.0}

                def timed(P, solve=solve, cnt=cnt):
                    t0 = time.perf_counter()
                    r = solve(P)
                    cnt["t"] += time.perf_counter() - t0
                    cnt["calls"] += 1
                    return r
                best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, timed, maxnodes=700)
                row["cpu_baseline_numpy"] = {"kind": "port", "node_solves_per_sec": cnt["calls"] / max(cnt["t"], 1e-9), "node_solves": cnt["calls"],
                                             "optimum": best, "sample": "the whole tree (at most 700 nodes), cold starts, oracle/ipm_ref.py on numpy as node solver"}
            except Exception as e:
                row["cpu_baseline_numpy"] = {"error": repr(e)}
            try:
                # the COMPILED figure: the same iteration in plain C, one thread, working on the nonzeros (oracle/cpu_ref.c) - what a
                # CPU backend in the style of the reference's (DSDP / SDPA are compiled code) does on one host core
                st = {}
                solve_c = bnb.cpu_c_node_solver(1e-6, st)
                best, y, nodes, failed = bnb.branch_and_bound(prob, inst.intvars, solve_c, maxnodes=700)
                row["cpu_baseline"] = {"kind": "own C restatement", "cores": 1, "node_solves_per_sec": st.get("calls", 0) / max(st.get("seconds", 0.0), 1e-9),
                                       "node_solves": st.get("calls", 0), "ms_per_ipm_iteration": 1e3 * st.get("seconds", 0.0) / max(1, st.get("iters", 0)),
                                       "optimum": best, "unresolved_nodes": failed,
                                       "sample": "the whole tree (at most 700 nodes), cold starts, oracle/cpu_ref.c (gcc -O3, no BLAS) as node "
                                                 "solver; time inside the C solve only"}
                gpu_cold = row.get("cold", {}).get("ms_per_ipm_iteration")
                if gpu_cold:
                    row["cpu_baseline"]["device_over_cpu_iteration_time"] = gpu_cold / max(row["cpu_baseline"]["ms_per_ipm_iteration"], 1e-12)
            except Exception as e:
                row["cpu_baseline"] = {"error": repr(e)}
        out[name.split(".")[0]] = row
    return out


def bench_bnb_sizes(hb):
    """ms per interior-point iteration of ONE node solve on both paths of the engine - one-launch kernel (csrc/solve1.hip) and general
    path (HIPSDP_SOLVE1=0) - over the sizes the one-launch kernel is offered: sparse variable matrices (three nonzeros per matrix and
    block), dense constant matrices, LP rows of density 0.3 (synthetic, seeded), and the root node of example_MkP.  Best of three solves,
    engine time / iterations.  The line says per case whether the two paths took the same number of iterations to the same objective
    (same_iterations_and_objective: on well-conditioned problems they do; 4 of 400 random shapes with cond(M) about 1e14 end differently,
    profiles/r04_c_solve1_fuzz.txt - a cold solve the kernel gives up on is solved again by the general path)."""
    import numpy as np
    for d in ("tests", os.path.join("tests", "harness")):
        if os.path.join(ROOT, d) not in sys.path:
            sys.path.insert(0, os.path.join(ROOT, d))
    import sdpa_io
    from core_container import CoreContainer      # (a container, no arithmetic: nothing of oracle/ takes part in the timed solves)

    def core_of(sizes, m, q, seed):
        rng = np.random.default_rng(seed)
        ystar = rng.standard_normal(m)
        blocks = []
        for n in sizes:
            A = np.zeros((m + 1, n, n))
            for i in range(1, m + 1):
                for _ in range(3):
                    r, c = rng.integers(0, n, 2)
                    v = rng.standard_normal()
                    A[i, r, c] += v
                    if r != c:
                        A[i, c, r] += v
            Zs = rng.standard_normal((n, n)); Zs = Zs @ Zs.T + 0.5 * np.eye(n)
            A[0] = np.tensordot(ystar, A[1:], axes=(0, 0)) - Zs
            blocks.append(A)
        D = rng.standard_normal((q, m)) * (rng.random((q, m)) < 0.3)
        c = D @ ystar - rng.random(q) - 0.1
        b = sum(np.array([np.trace(A[i]) for i in range(1, m + 1)]) for A in blocks) + (D.T @ np.ones(q) if q else 0.0)
        return CoreContainer(b, blocks, D, c)

    cases = [("blocks %s, m %d, q %d" % (sz, m, q), core_of(sz, m, q, 5)) for sz, m, q in
             [([10], 37, 85), ([16], 40, 40), ([24], 40, 40), ([32], 48, 40), ([12, 12, 12], 40, 40), ([30, 30], 50, 20)]]
    inst = sdpa_io.read_sdpa(os.path.join(ROOT, "tests", "golden", "instances", "example_MkP.dat-s.gz"))
    D, c = sdpa_io.lp_dense(inst)
    cases.append(("example_MkP root (n 15, m 105, q 240)", CoreContainer(inst.obj, sdpa_io.dense_blocks(inst), D, c)))
    rows = []
    keep = os.environ.get("HIPSDP_SOLVE1")
    try:
        for name, core in cases:
            row = {"case": name}
            for label, path in (("one_launch", "1"), ("general", "0")):
                os.environ["HIPSDP_SOLVE1"] = path
                s = hb.Solver(0)
                s.load_core(core)
                best = None
                for _ in range(3):
                    info = s.solve(gaptol=1e-6, feastol=1e-6)
                    best = info.solve_seconds if best is None else min(best, info.solve_seconds)
                row[label] = {"ms_per_ipm_iteration": 1e3 * best / max(1, info.iterations), "iterations": info.iterations, "status": info.status,
                              "dobj": info.dobj, "took_one_launch_path": bool(s.solve_path())}
                s.close()
            row["same_iterations_and_objective"] = bool(row["one_launch"]["iterations"] == row["general"]["iterations"]
                                                        and abs(row["one_launch"]["dobj"] - row["general"]["dobj"]) <= 1e-7 * (1 + abs(row["general"]["dobj"])))
            rows.append(row)
    finally:
        if keep is None:
            os.environ.pop("HIPSDP_SOLVE1", None)
        else:
            os.environ["HIPSDP_SOLVE1"] = keep
    return rows


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if not in_launcher and args.gpus > 1:
        return launch_ranks(args, argv)            # parent: no GPU call has been made in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: rank %d: --gpus %d does not match WORLD_SIZE %d of the launcher; refusing to report a line whose "
                         "n_gpus would not be what was asked for\n" % (rank, args.gpus, world))
        return 2
    if args.launch_dry_run:
        return dry_run_rank(args)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.device_count() <= local_rank:
            sys.stderr.write("bench.py: rank %d: LOCAL_RANK %d but only %d GPU(s) visible\n" % (rank, local_rank, torch.cuda.device_count()))
            return 3
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl")
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)

    form = args.schur_form
    if form == "rows":
        os.environ["HIPSDP_SCHUR"] = "R"                  # read by the engine when it sizes its Schur workspace (csrc/ipm.hip)
    elif form in ("cols", "vars"):
        os.environ.pop("HIPSDP_SCHUR", None)
    if form == "vars":
        args.shard_matrices = "on"
    elif form in ("cols", "rows"):
        args.shard_matrices = "off"
    hb = load_binding()
    if hb.device_count() <= 0:
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible and hipsdp has no CPU path")
    import ctypes as C
    lib = hb.lib()
    n, m = args.n, args.m
    solver = hb.Solver(local_rank if world > 1 else 0)
    # N > 1: ONE node SDP, its Schur assembly sharded over the ranks (north_star); every rank holds the same instance - or,
    # when the replicated constraint matrices would not fit the device (n=4000, m=8000: 1 TB), only the matrices of its variables
    comm = None
    rccl_ranks = None
    if world > 1:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            assert lib.hipsdp_comm_unique_id(buf) == 0
            uid = torch.tensor(list(buf), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, src=0)
        raw = (C.c_ubyte * 128)(*uid.cpu().tolist())
        comm = C.c_void_p()
        rc = lib.hipsdp_comm_create(raw, rank, world, C.byref(comm))
        assert rc == 0, "hipsdp_comm_create failed: %s" % lib.hipsdp_last_error().decode()
        cnt, kind = C.c_int(0), C.c_int(-1)
        assert lib.hipsdp_comm_count(comm, C.byref(cnt), C.byref(kind)) == 0
        rccl_ranks = int(cnt.value)
        assert rccl_ranks == world and kind.value == 0, "RCCL reports %d ranks (kind %d), launcher %d" % (rccl_ranks, kind.value, world)
        assert lib.hipsdp_set_comm(solver.h, comm, rank, world) == 0
        assert lib.hipsdp_shard_matrices(solver.h, {"auto": -1, "on": 1, "off": 0}[args.shard_matrices]) == 0
    solver.set_shape(m, [n], 0)
    Xs, Zs, ys = planted_pair(n, m, args.seed)
    b = solver.gen_planted(n, m, args.seed, Xs, Zs, ys)
    opt = float(b @ ys)

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.solve(gaptol=1e-5, feastol=1e-5)
    if comm is not None:
        lib.hipsdp_comm_stats_enable(comm, 1)
        lib.hipsdp_comm_stats(comm, None, None, None, 1)
    barrier()
    t0 = time.perf_counter()
    infos = [solver.solve(gaptol=1e-5, feastol=1e-5) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    last = infos[-1]
    ok = all(i.status == 0 for i in infos) and abs(last.dobj - opt) <= 1e-5 * (1 + abs(opt))
    iters = sum(i.iterations for i in infos)
    sharded = bool(lib.hipsdp_matrices_sharded(solver.h))
    # measured beside the vendor figure: the shader frequency during the assemblies of one more (untimed) solve -
    # one-thread kernels read the clock counters right before and after every assembly - and the matrix peak of THIS device under a
    # 20 ms register-only MFMA loop, with the frequency the firmware grants under that load
    measured = {}
    if rank == 0:
        try:
            ghz = C.c_double(0.0)
            lib.hipsdp_set_clock_sampling(solver.h, 1)
            solver.solve(gaptol=1e-5, feastol=1e-5)
            lib.hipsdp_get_assembly_clock(solver.h, C.byref(ghz))
            lib.hipsdp_set_clock_sampling(solver.h, 0)
            tf, pg = C.c_double(0.0), C.c_double(0.0)
            hb.ulib().hipsdp_mfma_peak(C.c_int(local_rank if world > 1 else 0), C.c_double(20.0), C.byref(tf), C.byref(pg))
            measured = {"assembly_clock_ghz": ghz.value, "measured_peak_tflops": tf.value, "measured_peak_clock_ghz": pg.value}
        except Exception as e:                                     # pragma: no cover
            measured = {"error": repr(e)}
    roof = schur_summary(infos, n, m, world)
    pmc = PMC_FROM_PROFILES.get((n, m)) if world == 1 else None
    roofline = {"bound": "mfma", "achieved": roof["achieved"], "peak": FP64_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                "frac": roof["frac"],
                # the same achieved rate against what this device delivers under pure matrix load, and against the vendor peak scaled
                # to the frequency the assemblies actually ran at (peak is quoted at 2.4 GHz)
                "measured_peak": measured.get("measured_peak_tflops"),
                "measured_peak_clock_ghz": measured.get("measured_peak_clock_ghz"),
                "frac_of_measured_peak": (roof["achieved"] / (measured["measured_peak_tflops"] * world)) if measured.get("measured_peak_tflops") else None,
                "assembly_clock_ghz": measured.get("assembly_clock_ghz"),
                "frac_of_peak_at_assembly_clock": (roof["achieved"] / (FP64_MFMA_PEAK_TFLOPS * world * measured["assembly_clock_ghz"] / 2.4))
                                                  if measured.get("assembly_clock_ghz") else None,
                "frac_is": "EXECUTED FP64 matrix-core flops of the Schur assembly (counted by the engine from the tiles and K ranges its "
                           "GEMM launches walk; live in this run) / HIP-event time of the assemblies / FP64 matrix peak - the quantity "
                           "MFMA-busy counters measure.  survey_8d_count_over_time_tflops divides SURVEY.md 8(d)'s count "
                           "4 m1 n^3 + m1^2 n^2 per assembly by the same time: NOT a rate of this hardware (the W formulation executes about "
                           "two thirds of that count), so it can exceed the peak and is no roofline fraction",
                "survey_8d_count_over_time_tflops": roof["algorithmic_equivalent_tflops"],
                "traffic": pmc["traffic_bytes_per_assembly"] if pmc else None,
                "traffic_is": ("quoted from the committed PMC profile, not measured in this run: " + pmc["source"]) if pmc else
                              "not measured for this size / rank count",
                "pmc_executed_tflops": pmc["executed_tflops"] if pmc else None,
                "pmc_mfma_busy": pmc["mfma_busy"] if pmc else None,
                "kernel": "Schur assembly: hs_dgemm5_kernel<1,1> / <1,2> (the two triangular n^3 products, paired-band kernel), hs_gram_kernel (Gram "
                          "product; hs_dgemm2_kernel<0,0> for shapes with more than one item per workgroup) + summation of the partial tiles; "
                          "the first assembly of a cold solve is the Gram product alone",
                "executed_flops_per_assembly": roof["executed_flops_per_assembly"],
                "algorithmic_flops_per_assembly": roof["algorithmic_flops_per_assembly"],
                "avg_assembly_ms": roof["avg_assembly_ms"], "assemblies": roof["assemblies"],
                "schur_share_of_solve_time": roof["schur_share_of_solve_time"]}
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
        "config": {"workload": "%s: synthetic single dense block n=%d, m=%d, fp64, planted optimum, cold start, gaptol=feastol=1e-5, "
                               "A resident in HBM" % (workload_name(n, m), n, m),
                   "parallelism": "1 GPU" if world == 1 else
                                  "one node SDP over %d ranks: Schur assembly sharded (%s), passes over A by rows (RCCL all-gather / "
                                  "all-reduce), everything else replicated" % (world, "vars: W_j formed where A_j lives, all-to-all of the W "
                                  "entries, all-reduce of the partial Schur matrices" if sharded else
                                  ("rows: row chunks of the Schur matrix in the U formulation, RCCL all-gather" if form == "rows" else
                                   "cols: column slices of W_j = G A_j R, RCCL all-reduce of the partial Schur matrices")),
                   "schur_form": "vars" if sharded else ("rows" if form == "rows" else "cols"),
                   "matrices": "sharded by variable" if sharded else "replicated",
                   "n": n, "m": m, "seed": args.seed},
        "iters_per_sec": iters / elapsed,
        "iterations_per_solve": iters / max(1, len(infos)),
        "solution_check": {"status_optimal_and_objective_matches_planted_optimum": bool(ok), "objective": last.dobj,
                           "planted_optimum": opt, "pinf": last.pinf, "dabs": last.dabs, "gap": last.gap},
        "roofline": roofline,
    }
    if comm is not None:
        sec, calls, byt = (C.c_double * 4)(), (C.c_longlong * 4)(), (C.c_double * 4)()
        lib.hipsdp_comm_stats(comm, sec, calls, byt, 1)
        lib.hipsdp_comm_stats_enable(comm, 0)
        names = ["schur_exchange", "passes_over_A", "decision_scalars", "other"]
        out["rccl_ranks"] = rccl_ranks
        out["collectives"] = {"unit": "ms per solve on rank 0 (device time between HIP events around each RCCL call)",
                              **{names[p]: {"ms_per_solve": 1e3 * sec[p] / args.steps, "calls_per_solve": calls[p] / args.steps,
                                            "MB_per_solve": byt[p] / args.steps / 1e6} for p in range(4)}}
    if rank == 0 and world == 1 and not args.no_extras and ok:
        try:
            out["phases"] = phase_anatomy(hb, solver)
        except Exception as e:                                   # the headline number stands on its own
            out["phases"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_extras and ok:
        # the same solves with the first assembly through the general products: at the cold start X = Z = xi I the Schur complement is
        # the Gram matrix of the constraint matrices themselves (M_ij = <A_i, A_j>), so the headline run does not multiply by
        # sqrt(xi) I and I / sqrt(xi) in its first iteration (csrc/schur.hip: hs_schur_W_identity); this is what that is worth
        try:
            os.environ["HIPSDP_NO_IDENTITY_START"] = "1"
            nrep = max(3, min(10, args.steps))
            solver.solve(gaptol=1e-5, feastol=1e-5)
            barrier()
            t0g = time.perf_counter()
            gi = [solver.solve(gaptol=1e-5, feastol=1e-5) for _ in range(nrep)]
            barrier()
            elg = time.perf_counter() - t0g
            out["cold_start_first_assembly"] = {
                "what": "value counts cold solves whose FIRST Schur assembly is the Gram product alone (X = Z = xi I at the cold start: "
                        "M_ij = tr(A_i X A_j Z^-1) = <A_i, A_j>; nothing is skipped that the result depends on - the objective of the two runs "
                        "agrees to rounding); 'general_first_assembly' runs the two n^3 products in that iteration as well "
                        "(HIPSDP_NO_IDENTITY_START=1), same process, same data",
                "general_first_assembly_solves_per_sec": nrep / elg,
                "general_first_assembly_ms_per_step": 1e3 * elg / nrep,
                "general_first_assembly_objective": gi[-1].dobj, "headline_objective": last.dobj,
                "same_iterations": bool(all(g.iterations == last.iterations for g in gi))}
        except Exception as e:
            out["cold_start_first_assembly"] = {"error": repr(e)}
        finally:
            os.environ.pop("HIPSDP_NO_IDENTITY_START", None)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(solver, b, n, m, int(round(iters / max(1, len(infos)))))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_extras and ok and (n, m) == (500, 1000):
        for key, fn in (("warm_start", lambda: bench_warm_start(solver, n, m, b, barrier, last.iterations)),
                        ("sdpi_boundary", lambda: bench_sdpi_boundary(hb, solver, n, m, b, opt))):
            try:
                out[key] = fn()
            except Exception as e:
                out[key] = {"error": repr(e)}
    if comm is not None:
        lib.hipsdp_set_comm(solver.h, None, 0, 1)
    solver.close()
    if comm is not None:
        lib.hipsdp_comm_destroy(comm)
    if rank == 0 and world == 1 and not args.no_extras and ok and (n, m) == (500, 1000):
        try:
            out["t1"] = bench_t1(hb, args.seed, barrier, cpu=not args.no_cpu)
        except Exception as e:
            out["t1"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_extras and ok and (n, m) == (500, 1000):
        try:
            out["density"] = bench_density(hb, n, m, args.seed, args.density, barrier)
        except Exception as e:
            out["density"] = {"error": repr(e)}
        try:
            out["bnb"] = bench_bnb(hb, cpu=not args.no_cpu)
            try:
                out["bnb"]["one_launch_against_general_path"] = bench_bnb_sizes(hb)
            except Exception as e:
                out["bnb"]["one_launch_against_general_path"] = {"error": repr(e)}
        except Exception as e:
            out["bnb"] = {"error": repr(e)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    return 0 if ok else 2


if __name__ == "__main__":
    sys.exit(main())
