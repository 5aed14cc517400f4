"""Sharded node solve with several PROCESSES (gpu-marked): WORLD ranks on the one device of the test box, joined by the
host-staged communicator (RCCL refuses two ranks per device), must give the solve a single rank gives.  This runs exactly the
code the multi-GPU bench runs per rank - row-sharded Schur assembly, two all-gathers, re-assembly, broadcast of the decision
scalars - with a different byte transport underneath."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_world(tmp_path, world, n, m, q, tag, env=None, load="dense", n2=0):
    name = "/hipsdp_t_%d_%s" % (os.getpid(), tag)
    outs = [str(tmp_path / ("%s_r%d.json" % (tag, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multi_worker.py"), str(r), str(world), name, str(n), str(m),
                               str(q), outs[r], load, str(n2)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, **(env or {}))) for r in range(world)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=280)
            logs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("form", ["columns", "rows", "columns+passes", "columns, small workspace"])
@pytest.mark.parametrize("world,n,m,q", [(2, 40, 70, 0), (3, 33, 50, 7), (4, 70, 130, 0), (2, 24, 1, 0), (4, 20, 5, 3)])
def test_ranks_sharing_the_schur_matrix_reproduce_the_single_rank_solve(gpu, tmp_path, world, n, m, q, form):
    """form = columns: the default (column slices of the W formulation, all-reduce); rows: HIPSDP_SCHUR=R (row chunks of the U
    formulation, two all-gathers); columns+passes: additionally every pass over the constraint matrices is swept by rows
    (each rank its ceil(m1 / ranks) matrices; all-gather of A(V), all-reduce of the packed A^T(coef) - forced here, by size in
    production).  n = 20 and 33 leave ranks without a column, m = 1 and 5 leave ranks without a row: they contribute zeros."""
    one = run_world(tmp_path, 1, n, m, q, "w1")[0]
    env = {"HIPSDP_SCHUR": "R"} if form == "rows" else ({"HIPSDP_SHARD_PASSES": "1"} if form == "columns+passes" else {"HIPSDP_SHARD_PASSES": "0"})
    if form == "columns, small workspace":
        env["HIPSDP_WS_GB"] = "0.0004"          # 400 kB: several column slices per rank (or the row form when even 16 columns do not fit)
    many = run_world(tmp_path, world, n, m, q, "w%d" % world, env=env)
    assert one["status"] == 0
    y1 = np.array(one["y"])
    for r in many:
        assert r["status"] == 0 and r["iterations"] == one["iterations"], (r["status"], r["iterations"], one["iterations"])
        assert np.max(np.abs(np.array(r["y"]) - y1)) <= 1e-8 * max(1.0, np.max(np.abs(y1)))
        assert abs(r["xtrace"] - one["xtrace"]) <= 1e-7 * max(1.0, abs(one["xtrace"]))
    # all ranks hold the same iterate (the decision scalars are broadcast, everything else is replicated arithmetic)
    for r in many[1:]:
        assert r["iterations"] == many[0]["iterations"]
        assert np.max(np.abs(np.array(r["y"]) - np.array(many[0]["y"]))) <= 1e-12 * max(1.0, np.max(np.abs(y1)))


@pytest.mark.parametrize("load,slice_cols,staging", [("vars-dense", 0, 1 << 20), ("vars-coo", 16, 1 << 20), ("vars-gen", 0, 1 << 20),
                                                     ("vars-dense", 24, 4096)])
@pytest.mark.parametrize("world,n,m,q", [(2, 70, 130, 0), (3, 80, 50, 7), (4, 96, 40, 0), (4, 20, 5, 3), (2, 24, 1, 0), (3, 2, 2, 0)])
def test_matrices_sharded_by_variable_reproduce_the_single_rank_solve(gpu, tmp_path, world, n, m, q, load, slice_cols, staging):
    """hipsdp_shard_matrices(1): every rank holds only the matrices of its ceil(m1 / ranks) variables (the
    layout n = 4000 / m = 8000 needs).  W_j = G A_j R is formed where A_j lives, the entries of the W_j are re-distributed by an
    all-to-all per column slice, the partial Gram matrices are summed; the passes over A are swept by rows.  The three loaders
    (dense copy, COO scatter, device-side generator) store only the rows a rank holds.  slice_cols forces several column slices
    (several all-to-alls per assembly), the 4 kB staging segment cuts every all-to-all into pieces; m = 1 and 5 leave ranks
    without a matrix, n = 2 leaves a rank without a row of the W_j."""
    if load == "vars-gen" and q > 0:
        pytest.skip("the device-side generator has no LP rows")
    one = run_world(tmp_path, 1, n, m, q, "v1", load=load.split("-")[1])[0]
    env = {"HIPSDP_TEST_STAGING": str(staging)}
    if slice_cols:
        env["HIPSDP_VAR_SLICE"] = str(slice_cols)
    many = run_world(tmp_path, world, n, m, q, "v%d" % world, env=env, load=load)
    assert one["status"] == 0
    y1 = np.array(one["y"])
    for r in many:
        assert r["status"] == 0 and r["iterations"] == one["iterations"], (r["status"], r["iterations"], one["iterations"])
        assert np.max(np.abs(np.array(r["y"]) - y1)) <= 1e-8 * max(1.0, np.max(np.abs(y1)))
        assert abs(r["xtrace"] - one["xtrace"]) <= 1e-7 * max(1.0, abs(one["xtrace"]))
    for r in many[1:]:
        assert np.max(np.abs(np.array(r["y"]) - np.array(many[0]["y"]))) <= 1e-12 * max(1.0, np.max(np.abs(y1)))


@pytest.mark.parametrize("world,n,m,q,slice_cols", [(2, 70, 130, 0, 16), (3, 80, 50, 7, 24), (4, 96, 40, 0, 32)])
def test_overlapped_exchange_of_the_variable_sharded_assembly_is_bit_identical_to_the_in_order_form(gpu, tmp_path, world, n, m, q, slice_cols):
    """several column slices per assembly: by default the all-to-all of slice s runs on the communication queue behind an event while
    the compute queue forms the products of slice s + 1, and the Gram update of slice s waits for its exchange (csrc/schur.hip:
    hs_schur_Wvar_all; SURVEY.md section 7.3: the exchange "must be overlapped").  HIPSDP_VAR_OVERLAP=0 issues everything in order
    on one queue.  Same kernels, same arguments, same order of the updates: the two runs must agree in every bit on every rank."""
    env = {"HIPSDP_TEST_STAGING": str(1 << 20), "HIPSDP_VAR_SLICE": str(slice_cols)}
    over = run_world(tmp_path, world, n, m, q, "ov%d" % world, env=dict(env, HIPSDP_VAR_OVERLAP="1"), load="vars-dense")
    inord = run_world(tmp_path, world, n, m, q, "io%d" % world, env=dict(env, HIPSDP_VAR_OVERLAP="0"), load="vars-dense")
    one = run_world(tmp_path, 1, n, m, q, "o1", load="dense")[0]
    for a, b in zip(over, inord):
        assert a["status"] == b["status"] == 0 and a["iterations"] == b["iterations"] == one["iterations"]
        assert np.array_equal(np.array(a["y"]), np.array(b["y"]))
        assert a["xtrace"] == b["xtrace"]
    y1 = np.array(one["y"])
    assert np.max(np.abs(np.array(over[0]["y"]) - y1)) <= 1e-8 * max(1.0, np.max(np.abs(y1)))


def test_matrices_sharded_by_variable_at_a_size_that_uses_the_mfma_tile_kernels(gpu, tmp_path):
    """n = 300, m = 400 on two ranks: the three products of hs_schur_Wvar run on the persistent 128-tile FP64-MFMA kernel, the
    instance is generated on the device by rows, the packed copies hold a rank's rows only"""
    one = run_world(tmp_path, 1, 300, 400, 0, "b1", load="gen")[0]
    many = run_world(tmp_path, 2, 300, 400, 0, "b2", env={"HIPSDP_TEST_STAGING": str(1 << 26), "HIPSDP_VAR_SLICE": "160"}, load="vars-gen")
    assert one["status"] == 0
    y1 = np.array(one["y"])
    for r in many:
        assert r["status"] == 0 and r["iterations"] == one["iterations"]
        assert np.max(np.abs(np.array(r["y"]) - y1)) <= 1e-8 * max(1.0, np.max(np.abs(y1)))


@pytest.mark.parametrize("name,world,shard_small", [("example_CLS.dat-s.gz", 2, True), ("example_small.dat-s", 3, True),
                                              ("example_CLS.dat-s.gz", 2, False), ("example_TT.dat-s.gz", 2, False)])
def test_spmd_copies_of_a_branch_and_bound_run_over_the_solver_interface(gpu, tmp_path, name, world, shard_small):
    """N identical processes run the same branch-and-bound over SCIPsdpiSolverLoadAndSolve (tests/spmd_worker.py knows nothing
    about ranks); HIPSDP_WORLD / HIPSDP_RANK / HIPSDP_COMM_SHM make sdpisolver_hip.c join the process-wide communicator, so every
    node SDP is sharded over the ranks.  All copies must walk the same tree to the same optimum as a single process.
    shard_small = False is the production rule: node SDPs this small are solved by every copy on its own with the single-launch
    kernels and rank 0's outcome is broadcast once per solve (example_TT: 569 nodes)."""
    def run(n, tag):
        outs = [str(tmp_path / ("%s_%d.json" % (tag, r))) for r in range(n)]
        env = dict(os.environ)
        if n > 1:
            env.update(HIPSDP_WORLD=str(n), HIPSDP_COMM_SHM="/hipsdp_spmd_%d_%s" % (os.getpid(), tag), HIPSDP_COMM_TIMEOUT="60")
            if shard_small:
                env["HIPSDP_SHARD_MIN_FLOPS"] = "0"      # shard even these tiny node SDPs (default: small problems run replicated)
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "spmd_worker.py"), name, outs[r]], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, env=dict(env, HIPSDP_RANK=str(r))) for r in range(n)]
        logs = []
        try:
            for p in procs:
                o, _ = p.communicate(timeout=280)
                logs.append(o.decode(errors="replace"))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        for r, p in enumerate(procs):
            assert p.returncode == 0, "copy %d failed:\n%s" % (r, logs[r][-3000:])
        return [json.load(open(o)) for o in outs]

    one = run(1, "s1")[0]
    many = run(world, "s%d" % world)
    assert one["best"] is not None      # example_small has nodes the plain call cannot settle (sdpi.c answers them with the penalty form)
    for r in many:
        # the copies agree with each other to the last bit: same tree, same calls, same iterates
        assert (r["failed"], r["nodes"], r["calls"], r["iters"]) == (many[0]["failed"], many[0]["nodes"], many[0]["calls"], many[0]["iters"])
        assert r["y"] == many[0]["y"]
        assert abs(r["best"] - one["best"]) <= 1e-6 * max(1.0, abs(one["best"]))
        if shard_small:
            # the general kernels replace the single-launch small ones: rounding differs, and the few nodes of example_small whose
            # optimum is not attained (tau -> 0) may be settled on another rung of the settings ladder or be branched on instead -
            # the tree may differ by such nodes, the optimum may not
            assert r["failed"] <= 1 and abs(r["nodes"] - one["nodes"]) <= 4, (r["nodes"], one["nodes"], r["failed"])
        else:
            assert (r["failed"], r["nodes"], r["calls"], r["iters"]) == (one["failed"], one["nodes"], one["calls"], one["iters"])


@pytest.mark.parametrize("slow", [1, 0])
def test_time_limit_decision_is_collective_when_the_clocks_of_the_copies_disagree(gpu, tmp_path, slow):
    """Two SPMD copies call SCIPsdpiSolverLoadAndSolve with the same time limit (0.5 s) while ONE copy's clock is already 1 s old.
    Each process has its own SDPIclock; a copy that returned early on its own reading would leave the other one alone in the next
    RCCL collective, which has no timeout.  The backend takes rank 0's reading on every copy: slow = 1 -> rank 0 still has time,
    both solve (identical results); slow = 0 -> rank 0 is out of time, both report the time limit without touching the engine.
    A second call with a 1 ms limit (expired on both) must report the time limit on both."""
    outs = [str(tmp_path / ("tl_%d.json" % r)) for r in range(2)]
    env = dict(os.environ)
    env.update(HIPSDP_WORLD="2", HIPSDP_COMM_SHM="/hipsdp_tl_%d_%d" % (os.getpid(), slow), HIPSDP_COMM_TIMEOUT="40",
               HIPSDP_SHARD_MIN_FLOPS="0", SLOW_RANK=str(slow), SLOW_SECONDS="1.0", TIME_LIMIT="0.5")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "spmd_timelimit_worker.py"), outs[r]], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, env=dict(env, HIPSDP_RANK=str(r))) for r in range(2)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=150)
            logs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "copy %d failed:\n%s" % (r, logs[r][-3000:])
    res = [json.load(open(o)) for o in outs]
    first = [r[0] for r in res]
    if slow == 1:
        assert all(f["rc"] == 1 and f["solved"] and f["optimal"] and not f["timelim"] for f in first), first
        assert first[0]["obj"] == first[1]["obj"]
        assert first[1]["clock"] > 0.5 > 0.0                    # the slow copy really was past the limit
    else:
        assert all(f["rc"] == 1 and not f["solved"] and f["timelim"] for f in first), first
    assert all(r[1]["rc"] == 1 and r[1]["timelim"] and not r[1]["solved"] for r in res), [r[1] for r in res]


@pytest.mark.parametrize("load", ["gen", "vars-gen"])
def test_two_ranks_at_the_bench_size(gpu, tmp_path, load):
    """BASELINE configs[1] (n = 500, m = 1000, the instance generated on the device) solved by two processes: replicated matrices
    (column slices + all-reduce, passes swept by rows) and matrices sharded by variable (all-to-all of the W entries) must give
    the single-process solve - this is bench.py --gpus 2 with the host-staged transport in place of RCCL"""
    one = run_world(tmp_path, 1, 500, 1000, 0, "c1", load="gen")[0]
    many = run_world(tmp_path, 2, 500, 1000, 0, "c2", env={"HIPSDP_TEST_STAGING": str(64 << 20)}, load=load)
    assert one["status"] == 0
    y1 = np.array(one["y"])
    for r in many:
        assert r["status"] == 0 and r["iterations"] == one["iterations"]
        assert np.max(np.abs(np.array(r["y"]) - y1)) <= 1e-10 * max(1.0, np.max(np.abs(y1)))


@pytest.mark.parametrize("load,env", [("dense", {}), ("dense", {"HIPSDP_SCHUR": "R"}), ("dense", {"HIPSDP_SHARD_PASSES": "1", "HIPSDP_WS_GB": "0.0004"}),
                                      ("vars-dense", {"HIPSDP_VAR_SLICE": "32"}), ("vars-coo", {})])
@pytest.mark.parametrize("world,n,n2,m,q", [(2, 70, 40, 90, 0), (3, 96, 130, 60, 5), (2, 200, 129, 260, 30)])
def test_two_blocks_of_different_sizes_on_several_ranks(gpu, tmp_path, world, n, n2, m, q, load, env):
    """two dense blocks sharing the Schur workspace, the split-K slabs and the communicator buffers: column slices, row chunks,
    row-swept passes with a small workspace, and matrices sharded by variable (dense and COO loaders) against the single process"""
    one = run_world(tmp_path, 1, n, m, q, "t1", load=load.split("-")[-1], n2=n2)[0]
    many = run_world(tmp_path, world, n, m, q, "t%d" % world, env=dict(env, HIPSDP_TEST_STAGING=str(1 << 22)), load=load, n2=n2)
    assert one["status"] == 0
    y1 = np.array(one["y"])
    for r in many:
        assert r["status"] == 0 and r["iterations"] == one["iterations"], (r["status"], r["iterations"], one["iterations"])
        assert np.max(np.abs(np.array(r["y"]) - y1)) <= 1e-8 * max(1.0, np.max(np.abs(y1)))


@pytest.mark.parametrize("shard_small", [True, False])
def test_time_limit_is_decided_by_rank_zero(gpu, tmp_path, shard_small):
    """the clocks of the ranks differ, so rank 0's time-limit decision is broadcast (sharded solve: every iteration; replicated
    small solve: with the outcome): all ranks stop at the same iteration with the same status, whatever their own clock says"""
    env = {"HIPSDP_TEST_TIMELIMIT": "0.003", "HIPSDP_SHARD_MIN_FLOPS": "0" if shard_small else "2e10"}
    many = run_world(tmp_path, 3, 70, 130, 0, "tl%d" % int(shard_small), env=env)
    assert many[0]["status"] in (0, 6)                       # optimal if the box was fast enough, else HIPSDP_STATUS_TIMELIM
    for r in many[1:]:
        assert r["status"] == many[0]["status"] and r["iterations"] == many[0]["iterations"]
