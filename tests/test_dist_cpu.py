"""world_size-2 gloo test (CPU) of the multi-GPU path: the Schur row sharding, the two all-gathers and the re-assembly,
restated in numpy (oracle/shard_ref.py), must reproduce the single-process Schur complement.  The RCCL call itself runs only
on the multi-GPU bench; the layout logic it relies on is what this test pins."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ipm_ref
import shard_ref


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m1, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)                         # every rank holds the same (replicated) data
    A = rng.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    G = rng.standard_normal((n, n)); X = G @ G.T + np.eye(n)
    G = rng.standard_normal((n, n)); Zi = np.linalg.inv(G @ G.T + np.eye(n))
    c, b1, b2, first, second = shard_ref.local_contribution(A, X, Zi, world, rank)

    def gather(block):
        pad = np.zeros((c, m1))
        pad[:block.shape[0]] = block
        out = [torch.zeros(c, m1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(out, torch.from_numpy(pad))
        return [o.numpy() for o in out]

    firsts = gather(first)
    seconds = gather(second)
    # trim the padded tails to the row counts the owners really have
    firsts = [f[:max(0, min(c, m1 - r * c))] for r, f in enumerate(firsts)]
    seconds = [s[:max(0, min(c, m1 - (2 * world - 1 - r) * c))] for r, s in enumerate(seconds)]
    Mx = shard_ref.assemble(m1, world, c, firsts, seconds)
    ref = ipm_ref.schur_block(A, X, Zi)
    err = float(np.abs(Mx - ref).max() / np.abs(ref).max())
    # per-rank work balance: number of upper-triangle entries computed
    work = sum(max(0, m1 - r) for r in list(range(b1, min(b1 + c, m1))) + list(range(b2, min(b2 + c, m1))))
    q.put((rank, err, work))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_schur_two_ranks():
    world, m1, n = 2, 23, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m1, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, work in res:
        assert err <= 1e-12, (rank, err)
    works = [w for _, _, w in res]
    assert max(works) - min(works) <= 2 * m1            # the paired chunks balance the triangle


def test_shard_rows_cover_everything_once():
    for m1 in (5, 23, 1001, 4001):
        for G in (1, 2, 4, 8):
            seen = np.zeros(m1 + 2 * G, dtype=int)
            for g in range(G):
                c, b1, b2 = shard_ref.shard_rows(m1, G, g)
                seen[b1:b1 + c] += 1
                seen[b2:b2 + c] += 1
            assert np.all(seen[:m1] == 1)


def _barrier_worker(rank, world, name, q):
    """create + destroy of the host-staged communicator needs no device: the creation ends in its first barrier"""
    import ctypes as C
    from conftest import _load_binding as load_binding
    lib = load_binding().lib()
    comm = C.c_void_p()
    rc = lib.hipsdp_comm_create_host(name.encode(), rank, world, C.c_longlong(1 << 16), C.c_double(30.0), C.byref(comm))
    if rc == 0:
        lib.hipsdp_comm_destroy(comm)
    q.put((rank, rc))


def test_host_staged_communicator_rendezvous():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/hipsdp_cpu_%d" % os.getpid()
    procs = [ctx.Process(target=_barrier_worker, args=(r, world, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, 0) for r in range(world)]
    assert not os.path.exists("/dev/shm" + name)            # rank 0 unlinks the name once everyone has mapped it


def test_host_staged_communicator_rejects_bad_arguments():
    import ctypes as C
    from conftest import _load_binding as load_binding
    lib = load_binding().lib()
    comm = C.c_void_p()
    assert lib.hipsdp_comm_create_host(b"no_slash", 0, 1, C.c_longlong(1 << 16), C.c_double(1.0), C.byref(comm)) != 0
    assert lib.hipsdp_comm_create_host(b"/x", 2, 2, C.c_longlong(1 << 16), C.c_double(1.0), C.byref(comm)) != 0
    assert lib.hipsdp_comm_create_host(b"/x", 0, 1, C.c_longlong(16), C.c_double(1.0), C.byref(comm)) != 0


def _cols_worker(rank, world, port, m1, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(12)
    A = rng.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    Gm = rng.standard_normal((n, n)); X = Gm @ Gm.T + np.eye(n)
    Gm = rng.standard_normal((n, n)); Z = Gm @ Gm.T + np.eye(n)
    R = np.linalg.cholesky(X)
    G = np.linalg.inv(np.linalg.cholesky(Z))
    bounds = shard_ref.shard_cols(m1, n, world)
    part = torch.from_numpy(shard_ref.column_slice_contribution(A, R, G, bounds[rank], bounds[rank + 1] - bounds[rank]))
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    ref = ipm_ref.schur_block(A, X, np.linalg.inv(Z))
    q.put((rank, float(np.abs(part.numpy() - ref).max() / np.abs(ref).max()), bounds))
    dist.barrier()
    dist.destroy_process_group()


def test_column_sliced_schur_two_ranks():
    world, m1, n = 2, 9, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cols_worker, args=(r, world, port, m1, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, bounds in res:
        assert err < 1e-12
        assert bounds[0] == 0 and bounds[-1] == n and all(b % 16 == 0 for b in bounds[:-1])


@pytest.mark.parametrize("m1,n,world", [(1001, 500, 1), (1001, 500, 2), (1001, 500, 8), (2001, 1000, 4), (2001, 1000, 8), (4001, 2000, 8),
                                        (31, 20, 4), (5, 7, 3), (200, 130, 16)])
def test_engine_column_partition_matches_the_restatement(m1, n, world):
    """the partition every rank derives on its own (host code of libhipsdp.so, no device needed) is the restated one: it
    tiles [0, n) without gaps, and no rank exceeds the others by more than the tile quantum allows"""
    import ctypes as C
    from conftest import _load_binding
    lib = _load_binding().lib()
    b = (C.c_int * (world + 1))()
    assert lib.hipsdp_shard_columns(m1, n, world, b) == 0
    bounds = list(b)
    assert bounds == shard_ref.shard_cols(m1, n, world)
    assert bounds[0] == 0 and bounds[-1] == n and all(x <= y for x, y in zip(bounds, bounds[1:]))
    assert lib.hipsdp_shard_columns(m1, n, 0, b) != 0


def _var_worker(rank, world, port, m1, n, cw, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(13)
    A = rng.standard_normal((m1, n, n)); A = A + A.transpose(0, 2, 1)
    Gm = rng.standard_normal((n, n)); X = Gm @ Gm.T + np.eye(n)
    Gm = rng.standard_normal((n, n)); Z = Gm @ Gm.T + np.eye(n)
    R = np.linalg.cholesky(X)
    G = np.linalg.inv(np.linalg.cholesky(Z))
    r0, r1 = shard_ref.var_rows(m1, world, rank)
    A_own = A[r0:r1].copy()                                  # the only matrices this rank touches
    q0, q1 = shard_ref.var_wrows(n, world, rank)
    part = torch.zeros(m1, m1, dtype=torch.float64)
    for c0 in range(0, n, cw):
        w = min(cw, n - c0)
        send = [torch.from_numpy(p) for p in shard_ref.var_send_pieces(A_own, R, G, world, c0, w)]
        recv = []
        for src in range(world):
            s0, s1 = shard_ref.var_rows(m1, world, src)
            recv.append(torch.zeros(s1 - s0, q1 - q0, w, dtype=torch.float64))
        # gloo has no all_to_all: the same exchange as pairwise sends and receives (RCCL: ncclSend / ncclRecv in one group)
        reqs = []
        for peer in range(world):
            if peer == rank:
                recv[rank].copy_(send[rank])
                continue
            if send[peer].numel() > 0:
                reqs.append(dist.isend(send[peer].contiguous(), dst=peer, tag=c0))
            if recv[peer].numel() > 0:
                reqs.append(dist.irecv(recv[peer], src=peer, tag=c0))
        for r in reqs:
            r.wait()
        part += torch.from_numpy(shard_ref.var_gram([t.numpy() for t in recv]))
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    ref = ipm_ref.schur_block(A, X, np.linalg.inv(Z))
    q.put((rank, float(np.abs(part.numpy() - ref).max() / np.abs(ref).max()), (r0, r1), (q0, q1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,m1,n,cw", [(2, 9, 12, 12), (3, 7, 10, 4), (2, 1, 5, 5)])
def test_variable_sharded_schur(world, m1, n, cw):
    """matrices sharded by variable: own W_j, exchange of row ranges, partial Gram matrices, all-reduce (restatement of
    hs_schur_Wvar); the row ranges cover everything once"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_var_worker, args=(r, world, port, m1, n, cw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, rows, wrows in res:
        assert err < 1e-12, (rank, err)
    assert res[0][2][0] == 0 and res[-1][2][1] == m1 and all(res[k][2][1] == res[k + 1][2][0] for k in range(world - 1))
    assert res[0][3][0] == 0 and res[-1][3][1] == n and all(res[k][3][1] == res[k + 1][3][0] for k in range(world - 1))
