"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (xGMI).

The reference has NO multi-device mechanism beyond "one host thread per device"
(docs/docs/start/architecture/multi-device.md:34-36,76); everything here is new design:

* MSM shards naturally -- sum of independent partial sums. The (scalar, base) pairs are split into
  world_size contiguous shards, every rank runs the full single-GPU MSM on its shard, and the
  world_size partial results (3*L words each: 96 B for bn254) are exchanged with ONE all_gather and
  summed on every rank with complete projective adds (bn254_hip_projective_sum). EC addition is not
  an RCCL reduce op, so "all-reduce of partial sums" is all_gather + local add; the payload is tiny,
  so xGMI bandwidth is irrelevant and the collective is latency-only.
* Batched NTT shards over rows (independent transforms): no data-path collective at all.

Both functions also run with the gloo backend on CPU tensors for the exchange step (the local
compute is injected), which is how tests/test_dist_cpu.py covers the N > 1 control flow without GPUs.
"""
import numpy as np


def shard_range(n: int, rank: int, world: int):
    """contiguous shard [lo, hi) of n items for `rank`; remainder spread over the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_partials(partial, world: int, dist, device=None):
    """partial: torch tensor [3*L] int32 (this rank's projective partial result). Returns [world, 3*L]."""
    import torch

    if world == 1:
        return partial.reshape(1, -1)
    out = torch.empty(world * partial.numel(), dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(out, partial.contiguous().reshape(-1))
    return out.reshape(world, -1)


def msm_sharded(curve: str, scalars_shard, bases_shard, n_shard: int, rank: int, world: int, dist, cfg=None):
    """This rank's shard is already resident (device pointers or torch tensors). Returns a torch int32
    tensor [3*L] on the rank's GPU holding the FULL result, identical on every rank."""
    import ctypes
    import torch
    from . import msm as M
    from ._lib import MSMConfig, lib, check

    L = M.LIMBS[curve]
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = cfg or MSMConfig.default()
    partial = torch.empty(3 * L, dtype=torch.int32, device=dev)
    sp = scalars_shard.data_ptr() if hasattr(scalars_shard, "data_ptr") else scalars_shard
    bp = bases_shard.data_ptr() if hasattr(bases_shard, "data_ptr") else bases_shard
    cfg.is_async = True
    M.msm(curve, sp, bp, cfg, results=partial.data_ptr(), msm_size=n_shard)
    if world == 1:
        return partial
    gathered = allgather_partials(partial, world, dist)
    out = torch.empty(3 * L, dtype=torch.int32, device=dev)
    check(getattr(lib, f"{curve}_hip_projective_sum")(gathered.data_ptr(), world, out.data_ptr(), None), "projective_sum")
    return out


def ntt_batch_shard(batch: int, rank: int, world: int):
    """rows [lo, hi) of a batch owned by `rank` (rows are independent transforms: no collective)."""
    return shard_range(batch, rank, world)


def combine_partials_host(curve: str, partials: np.ndarray):
    """Host-side definition of the combine step (used by the gloo CPU tests): sum of projective
    partials via the pure-Python oracle. partials: [world, 3*L] uint32."""
    from oracle import pyref

    C = pyref.CURVES[curve]
    L = C.limbs_q
    acc = pyref.INF
    for row in partials:
        x, y, z = (sum(int(v) << (32 * k) for k, v in enumerate(row[i * L:(i + 1) * L])) for i in range(3))
        acc = pyref.ec_add(C, acc, pyref.proj_to_affine(C, x, y, z))
    return acc
