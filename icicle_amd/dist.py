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


def _exchange_transpose(mat, world, dist):
    """mat: [R_local, C] with C divisible by world; rows are this rank's slice of a [R, C] matrix distributed
    by rows. Returns the [C/world, R] slice of the TRANSPOSE (row index = global column, in this rank's
    column slice). One all_to_all_single: every rank talks to all peers at once, which on xGMI's
    point-to-point links uses all 7 links instead of a ring's one."""
    import torch

    rl, c = mat.shape
    cl = c // world
    send = mat.reshape(rl, world, cl).permute(1, 0, 2).contiguous()  # [world][rl][cl]
    if world > 1:
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.reshape(-1), send.reshape(-1))
    else:
        recv = send
    return recv.permute(2, 0, 1).reshape(cl, world * rl).contiguous()


def ntt_distributed(field: str, chunk, logn: int, inverse: bool, rank: int, world: int, dist, natural_output: bool = True, compute=None):
    """ONE transform of size N = 2^logn whose input is distributed over `world` ranks in contiguous natural-order
    chunks (`chunk`: torch int32 [N/world]). 4-step (Bailey): N = N1*N2, j = j1*N2 + j2, k = k1 + N1*k2
      exchange  -> columns j2 on this rank        (all-to-all #1)
      N1-point NTTs over j1, twiddle w_N^(j2*k1)
      exchange  -> rows k1 on this rank           (all-to-all #2)
      N2-point NTTs over j2                       -> X[k1 + N1*k2] at [k1][k2]  ("mixed" order, kNM)
      exchange  -> natural chunks                 (all-to-all #3, only if natural_output)
    The NTT domain of `field` must be initialised with a root of order >= N on every rank.
    `compute` = (local_ntt(mat2d, inverse) -> mat2d, twiddle(mat2d, row0, logn, inverse) -> mat2d) lets the CPU
    gloo test inject the oracle; default = this backend on the current GPU."""
    import torch

    n = 1 << logn
    log_w = world.bit_length() - 1
    assert world == 1 << log_w and chunk.numel() == n // world
    a = max((logn + 1) // 2, log_w)
    b = logn - a
    assert b >= log_w, "transform too small for this many ranks"
    n1, n2 = 1 << a, 1 << b
    if compute is None:
        compute = _gpu_compute(field)
    local_ntt, twiddle = compute
    l0 = chunk.reshape(n1 // world, n2)
    t1 = _exchange_transpose(l0, world, dist)             # [n2/world][n1], row = global j2
    y = local_ntt(t1, inverse)                            # over j1 -> k1
    y = twiddle(y, rank * (n2 // world), logn, inverse)   # *= w_N^(+-j2*k1)
    t2 = _exchange_transpose(y, world, dist)              # [n1/world][n2], row = global k1
    z = local_ntt(t2, inverse)                            # over j2 -> k2 : X[k1 + n1*k2]
    if not natural_output:
        return z.reshape(-1)
    return _exchange_transpose(z, world, dist).reshape(-1)  # [n2/world][n1]: k = k1 + n1*k2 natural


def _gpu_compute(field: str):
    import ctypes
    import torch
    from . import ntt as N
    from ._lib import NTTConfigU32, lib, check

    def local_ntt(mat, inverse):
        rows, size = mat.shape
        out = torch.empty_like(mat)
        cfg = NTTConfigU32.default()
        cfg.batch_size = rows
        cfg.is_async = True
        N.ntt(field, mat.data_ptr(), N.INVERSE if inverse else N.FORWARD, cfg, out=out.data_ptr(), size=size)
        return out

    def twiddle(mat, row0, logn, inverse):
        rows, cols = mat.shape
        check(getattr(lib, f"{field}_hip_twiddle_rows")(mat.data_ptr(), rows, cols, row0, logn, inverse, None), "twiddle_rows")
        return mat

    return local_ntt, twiddle
