"""GPU: BASELINE configs[3] and configs[4] run WHOLE on the one MI355X of the test box, through the path a wrapper would
use for the 8-GPU job -- the unchanged msm() / ntt() symbols with config.ext {"hip_num_devices": 8} (8 shards; with one
visible GPU they are logical shards on its stream) -- and byte-compared with the reference CPU backend (VERDICT r02
item 1; rules: icicle/tests/test_curve_api.cpp:36-79 equality as group elements, test_mod_arithmetic_api.h:694 memcmp).

  configs[3]  BLS12-381 MSM, 2^28 scalars / bases (8 GiB + 24 GiB resident in HBM): E1 (partial sums) and E2 (bucket
              exchange) over 8 shards, the single-call 2^28 MSM, and the reference CPU backend on the same inputs
              (about five minutes on the box's 256 host cores: a background job of tests/refpool.py, started at session start).
  configs[4]  KoalaBear NTT 2^22 x 1024 (16 GiB): forward + inverse over 8 row shards, round trip and DC term on all
              1024 rows, memcmp with the reference on one row of every shard (oracle called on an 8-row batch).
"""
import ctypes

import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu


def _ext(**kv):
    from icicle_amd._lib import lib

    e = lib.create_config_extension()
    for k, v in kv.items():
        if isinstance(v, bool):
            lib.config_extension_set_bool(e, k.encode(), v)
        else:
            lib.config_extension_set_int(e, k.encode(), v)
    return e


def _config3_inputs(dev):
    """(scalars [2^28, 8], bases [2^28, 24]) int32 tensors on `dev`: uniform scalars below r, bases (2028 + i) G"""
    import torch
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check

    cname, logn, top = "bls12_381", 28, 0x73EDA753
    L = M.LIMBS[cname]
    n = 1 << logn
    bases = torch.empty((n, 2 * L), dtype=torch.int32, device=dev)
    check(getattr(lib, f"{cname}_hip_generate_affine_points")(bases.data_ptr(), n, 2028, True, None))
    g = torch.Generator(device=dev)
    g.manual_seed(28)
    sc = torch.empty((n, 8), dtype=torch.int32, device=dev)
    step = n // 8
    for k in range(8):  # (2^31 elements: drawn in slices)
        sc[k * step:(k + 1) * step] = torch.randint(-(2 ** 31), 2 ** 31, (step, 8), dtype=torch.int32, device=dev, generator=g)
        sc[k * step:(k + 1) * step, 7] = torch.randint(0, top, (step,), dtype=torch.int32, device=dev, generator=g)
    torch.cuda.synchronize()
    return sc, bases


def _job_config3(pool, hip, dev):
    """the reference CPU backend on the full 2^28 inputs (about five minutes on the box's 256 host cores): started at session
    start on a lane of its own (tests/refpool.py), joined by the test below at the end of the session"""
    sc, bases = _config3_inputs(dev)
    hs = np.ascontiguousarray(sc.cpu().numpy().view(np.uint32))
    del sc
    hb = np.ascontiguousarray(bases.cpu().numpy().view(np.uint32))
    del bases
    pool.submit_msm("config3", "bls12_381", hs, hb, lane="msm_big")


@pytest.mark.refjob("config3", order=100)
def test_config3_bls12_381_msm_2_28_whole(hip, refpool):
    import torch
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check

    cname, logn = "bls12_381", 28
    refc = ref.RefCurve(cname)
    L = M.LIMBS[cname]
    n = 1 << logn
    dev = torch.device("cuda", 0)
    sc, bases = _config3_inputs(dev)

    def run(ext=None):
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        out = np.zeros((1, 3 * L), dtype=np.uint32)
        M.msm(cname, sc.data_ptr(), bases.data_ptr(), cfg, results=out, msm_size=n)
        return out

    e1, e2 = _ext(hip_num_devices=8), _ext(hip_num_devices=8, hip_msm_exchange_buckets=True)
    try:
        r_e1, r_e2 = run(e1), run(e2)
    finally:
        lib.destroy_config_extension(e1)
        lib.destroy_config_extension(e2)
    single = run()
    aff = refc.to_affine(r_e1)
    assert refc.is_on_curve(r_e1[0]) and refc.is_on_curve(r_e2[0]) and refc.is_on_curve(single[0])
    assert np.array_equal(aff, refc.to_affine(r_e2)), "8 shards: bucket exchange differs from partial-sum exchange"
    assert np.array_equal(aff, refc.to_affine(single)), "8 shards differ from the single-call 2^28 MSM"
    check(lib.icicle_hip_release_workspace())
    del sc, bases
    torch.cuda.empty_cache()
    exp = refpool.result("config3")  # the reference CPU backend on the full 2^28 inputs
    assert np.array_equal(aff, refc.to_affine(exp)), "BLS12-381 2^28 over 8 shards: GPU result differs from the reference CPU backend"
    assert refc.projective_eq(r_e1[0], exp[0]) and refc.projective_eq(r_e2[0], exp[0])


_C4_PICK = [3, 130, 300, 400, 600, 700, 800, 1023]  # one row of every 128-row shard


def _config4_inputs(dev):
    import torch

    F = pyref.KOALABEAR
    logn, rows = 22, 1024
    n = 1 << logn
    g = torch.Generator(device=dev)
    g.manual_seed(44)
    x = torch.empty((rows, n), dtype=torch.int32, device=dev)
    for r0 in range(0, rows, 128):
        x[r0:r0 + 128] = torch.randint(0, F.p, (128, n), dtype=torch.int32, device=dev, generator=g)
    return x


def _job_config4(pool, hip, dev):
    """reference legs of config 4 (worker process, KoalaBear domain 2^22): the 8 sampled rows forward and back, and shard 5
    (rows 640..767) whole, forward and inverse"""
    x = _config4_inputs(dev)
    hx = np.ascontiguousarray(x[_C4_PICK].cpu().numpy().view(np.uint32)).reshape(-1)
    pool.submit_ntt("config4_pick", "koalabear", hx, 22, 0, batch=len(_C4_PICK), chain=[(1, 0, 1)], lane="ntt_a")
    hs = np.ascontiguousarray(x[640:768].cpu().numpy().view(np.uint32)).reshape(-1)
    pool.submit_ntt("config4_shard5_fwd", "koalabear", hs, 22, 0, batch=128, lane="ntt_a")
    pool.submit_ntt("config4_shard5_inv", "koalabear", hs, 22, 1, batch=128, lane="ntt_a")


@pytest.mark.refjob("config4_pick", "config4_shard5_fwd", "config4_shard5_inv", order=40)
def test_config4_koalabear_ntt_2_22_x_1024_whole(hip, refpool):
    import torch
    from icicle_amd import ntt as N
    from icicle_amd._lib import lib, check

    F = pyref.KOALABEAR
    logn, rows, G = 22, 1024, 8
    n = 1 << logn
    N.init_domain("koalabear", N.get_root_of_unity("koalabear", n))
    ext = _ext(hip_num_devices=G)
    try:
        dev = torch.device("cuda", 0)
        x = _config4_inputs(dev)
        y, z = torch.empty_like(x), torch.empty_like(x)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.is_async = rows, True
        cfg.ext = ext
        N.ntt("koalabear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
        N.ntt("koalabear", y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(x, z), "round trip over all 1024 rows"
        del z
        sums = torch.empty(rows, dtype=torch.int32, device=dev)  # X[0] of every row is the sum of the row
        for r0 in range(0, rows, 128):
            sums[r0:r0 + 128] = (x[r0:r0 + 128].to(torch.int64).sum(dim=1) % F.p).to(torch.int32)
        assert torch.equal(y[:, 0], sums)
        pick = _C4_PICK
        assert sorted({r // (rows // G) for r in pick}) == list(range(G))
        hx = np.ascontiguousarray(x[pick].cpu().numpy().view(np.uint32)).reshape(-1)
        hy = np.ascontiguousarray(y[pick].cpu().numpy().view(np.uint32)).reshape(-1)
        exp, back = refpool.result("config4_pick")  # reference forward of the sampled rows, and its inverse of that output
        assert np.array_equal(hy, exp), "forward 2^22 x 1024: sampled rows differ from the reference CPU backend"
        assert np.array_equal(back, hx)
        # one WHOLE 128-row shard (shard 5: rows 640..767) against the reference, forward and inverse
        es = refpool.result("config4_shard5_fwd")
        assert np.array_equal(np.ascontiguousarray(y[640:768].cpu().numpy().view(np.uint32)).reshape(-1), es), "forward, shard 5 whole"
        zi = torch.empty((128, n), dtype=torch.int32, device=dev)
        cfgs = hip.NTTConfigU32.default()
        cfgs.batch_size, cfgs.is_async = 128, True
        N.ntt("koalabear", x[640:768].data_ptr(), N.INVERSE, cfgs, out=zi.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert np.array_equal(np.ascontiguousarray(zi.cpu().numpy().view(np.uint32)).reshape(-1), refpool.result("config4_shard5_inv")), "inverse, shard 5 whole"
        del es, zi
        for k in ("config4_pick", "config4_shard5_fwd", "config4_shard5_inv"):
            refpool.drop(k)
        # the same shards without the extension (one launch sequence over all rows) give the same bytes
        y2 = torch.empty((256, n), dtype=torch.int32, device=dev)
        cfg2 = hip.NTTConfigU32.default()
        cfg2.batch_size, cfg2.is_async = 256, True
        N.ntt("koalabear", x[512:768].data_ptr(), N.FORWARD, cfg2, out=y2.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(y2, y[512:768])
        del x, y, y2
        torch.cuda.empty_cache()
    finally:
        lib.destroy_config_extension(ext)
        N.release_domain("koalabear")


# background reference jobs of this module: key -> (start priority, starter); tests/conftest.py runs the starters of the selected tests
REF_JOBS = {"config3": (0, _job_config3)}
for _k in ("config4_pick", "config4_shard5_fwd", "config4_shard5_inv"):
    REF_JOBS[_k] = (30, _job_config4)
