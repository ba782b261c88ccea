"""CPU, world_size 2, gloo: the N > 1 control flow of the sharded MSM (contiguous sharding, all_gather
of per-rank partial results, combine) and the batch sharding of the NTT. The per-rank compute is the
C oracle here (no GPU); on the GPU box the same exchange runs over RCCL with libicicle_hip.so doing
the compute (icicle_amd/dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icicle_amd import dist as D
    from oracle import pyref
    from tests import oracle_c as oc
    from tests.util import cached_points, combine_partials_host, points_to_array, rand_scalars, to_words

    C = pyref.BN254
    n = 101  # not divisible by world: exercises the remainder logic
    rng = np.random.default_rng(5)  # same seed on every rank -> same global problem
    pts = cached_points(C, n)
    sc = rand_scalars(rng, n, C.r)
    lo, hi = D.shard_range(n, rank, world)
    part = oc.msm("bn254", to_words(sc[lo:hi], 8), points_to_array(C, pts[lo:hi]), c=6)
    partial = torch.from_numpy(part.view(np.int32).copy())
    gathered = D.allgather_partials(partial, world, dist)
    full = combine_partials_host("bn254", gathered.numpy().view(np.uint32))
    exp = pyref.msm_naive(C, sc, pts)
    # batch sharding of the NTT: rows are disjoint and cover the batch
    rows = D.ntt_batch_shard(7, rank, world)
    q.put((rank, full == exp, (lo, hi), rows))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    (lo0, hi0), (lo1, hi1) = res[0][2], res[1][2]
    assert lo0 == 0 and hi0 == lo1 and hi1 == 101
    assert res[0][3] == (0, 4) and res[1][3] == (4, 7)


def _ntt_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icicle_amd import dist as D
    from oracle import pyref
    from tests import oracle_c as oc

    F = pyref.BABYBEAR
    ok = True
    for logn, inverse, natural in ((8, False, True), (9, True, True), (10, False, False)):
        n = 1 << logn
        root = pyref.omega(F, logn)
        rng = np.random.default_rng(100 + logn)  # same data on every rank
        x = rng.integers(0, F.p, size=n, dtype=np.uint32)
        exp = oc.ntt("babybear", x, n, root, inverse=inverse)

        def local_ntt(mat, inv):
            a = np.ascontiguousarray(mat.numpy().view(np.uint32))
            rows, size = a.shape
            dom = pyref.omega(F, size.bit_length() - 1)
            return torch.from_numpy(oc.ntt("babybear", a.reshape(-1), size, dom, inverse=inv, batch=rows).view(np.int32).reshape(rows, size).copy())

        def twiddle(mat, row0, ln, inv):
            a = mat.numpy().view(np.uint32).astype(object)
            w = root if not inv else pow(root, -1, F.p)
            rows, cols = a.shape
            for r in range(rows):
                for c in range(cols):
                    a[r, c] = int(a[r, c]) * pow(w, (row0 + r) * c, F.p) % F.p
            return torch.from_numpy(a.astype(np.uint32).view(np.int32).reshape(rows, cols).copy())

        lo, hi = D.shard_range(n, rank, world)
        chunk = torch.from_numpy(x[lo:hi].view(np.int32).copy())
        got = D.ntt_distributed("babybear", chunk, logn, inverse, rank, world, dist, natural_output=natural, compute=(local_ntt, twiddle))
        got = got.numpy().view(np.uint32)
        if natural:
            ok &= bool(np.array_equal(got, exp[lo:hi]))
        else:  # mixed order: this rank holds X[k1 + n1*k2] for its k1 slice, all k2
            a = max((logn + 1) // 2, 1)
            n1, n2 = 1 << a, 1 << (logn - a)
            k1 = np.arange(rank * n1 // world, (rank + 1) * n1 // world)
            ref = exp[(k1[:, None] + n1 * np.arange(n2)[None, :])].reshape(-1)
            ok &= bool(np.array_equal(got, ref))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_ntt_all_to_all_gloo_world2():
    """the 4-step transform split over ranks with all_to_all exchanges, oracle doing the local math"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ntt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_shard_range_properties():
    from icicle_amd.dist import shard_range

    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
