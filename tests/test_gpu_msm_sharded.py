"""GPU parity of the multi-GPU MSM path, rehearsed with G LOGICAL shards on one device (SURVEY.md 8(e) caveat):

* <curve>_hip_projective_sum (k_proj_sum, the combine step after the RCCL all-gather) on reference-generated
  projective points -- identity partials, duplicates and P + (-P) included -- against the reference's own ecadd /
  to_affine (comparison rule: icicle/tests/test_curve_api.cpp:77, equality as group elements);
* one MSM cut into G in {1, 2, 3, 8} contiguous shards exactly as icicle_amd/dist.py cuts it (shard_range), each
  shard's MSM on the GPU, the partial results combined by the device kernel, compared with the single-call MSM and
  with the reference CPU backend;
* dist.msm_sharded itself with world = 1 (the code path every rank runs before the exchange);
* the in-library multi-device MSM (config.ext "hip_num_devices" = G): the C-ABI caller's route to the same path.
"""
import ctypes

import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import cached_points, points_to_array, rand_scalars, to_words

pytestmark = pytest.mark.gpu
CURVES = ["bn254", "bls12_381"]


def _proj_sum(hip, cname, partials: np.ndarray, g2=False) -> np.ndarray:
    from icicle_amd._lib import lib, check
    from icicle_amd.runtime import DeviceVec

    sym = f"{cname}_g2" if g2 else cname
    partials = np.ascontiguousarray(partials, dtype=np.uint32)
    n, w = partials.shape
    d_in = DeviceVec.from_host(partials) if n else DeviceVec(16)
    d_out = DeviceVec(w * 4)
    check(getattr(lib, f"{sym}_hip_projective_sum")(d_in.ptr, n, d_out.ptr, None), "projective_sum")
    from icicle_amd import runtime

    runtime.device_synchronize()
    return d_out.to_host(shape=(1, w))


def _ref_sum(refc, rows: np.ndarray) -> np.ndarray:
    """fold with the reference's own projective add (<curve>_ecadd)"""
    L3 = rows.shape[1]
    acc = np.zeros(L3, dtype=np.uint32)
    acc[L3 // 3] = 1  # (0 : 1 : 0)
    fn = getattr(refc.lib, f"{refc.sym}_ecadd")
    for r in rows:
        out = np.zeros(L3, dtype=np.uint32)
        r = np.ascontiguousarray(r)
        fn(ctypes.c_void_p(acc.ctypes.data), ctypes.c_void_p(r.ctypes.data), ctypes.c_void_p(out.ctypes.data))
        acc = out
    return acc.reshape(1, L3)


@pytest.mark.parametrize("cname", CURVES)
def test_projective_sum_vs_reference_ecadd(hip, cname):
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(41)
    L = C.limbs_q
    # non-trivial projective representatives: results of small reference MSMs (Z != 1)
    rows = []
    for k in range(9):
        n = 5 + k
        bases = points_to_array(C, cached_points(C, 40)[k:k + n])
        rows.append(refc.msm(to_words(rand_scalars(rng, n, C.r), 8), bases)[0])
    rows = np.stack(rows)
    ident = np.zeros(3 * L, dtype=np.uint32)
    ident[L] = 1
    # negation of rows[2]: (x : -y : z)
    neg = rows[2].copy()
    y = sum(int(v) << (32 * i) for i, v in enumerate(neg[L:2 * L]))
    neg[L:2 * L] = to_words([(C.q - y) % C.q], L)[0]
    cases = {
        "one": rows[:1], "two": rows[:2], "nine": rows, "identity_first": np.vstack([ident, rows[:3]]),
        "identity_middle": np.vstack([rows[0], ident, rows[1], ident]), "all_identity": np.stack([ident] * 5),
        "duplicates": np.vstack([rows[4], rows[4], rows[4]]), "cancel": np.vstack([rows[2], neg]),
        "cancel_plus": np.vstack([rows[2], rows[5], neg]), "many": np.vstack([rows] * 15),  # 135 > 64 lanes
    }
    for name, part in cases.items():
        got = _proj_sum(hip, cname, part)
        exp = _ref_sum(refc, part)
        assert np.array_equal(refc.to_affine(got), refc.to_affine(exp)), (cname, name)
        assert refc.is_on_curve(got[0]) and refc.projective_eq(got[0], exp[0]), (cname, name)
        z = got[0][2 * L:]
        yv = got[0][L:2 * L]
        assert z.any() or yv.any(), "(0,0,0) is not a valid identity representative"
    # n = 0 -> identity
    got = _proj_sum(hip, cname, np.zeros((0, 3 * L), dtype=np.uint32))
    assert not got[0][:L].any() and got[0][L:2 * L].any() and not got[0][2 * L:].any()


@pytest.mark.parametrize("cname", CURVES)
@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_msm_logical_shards_on_one_gpu(hip, cname, G):
    """the N > 1 data path with G logical shards on GPU 0: shard -> msm -> (all-gather) -> k_proj_sum"""
    from icicle_amd import msm as M
    from icicle_amd.dist import shard_range

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(50 + G)
    n = 3001  # not divisible by 2, 3 or 8
    pts = list(cached_points(C, n))
    pts[7] = pyref.INF
    bases = points_to_array(C, pts)
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    if G == 8:
        sc[shard_range(n, 5, G)[0]:shard_range(n, 5, G)[1]] = 0  # one shard contributes the identity
    partials = []
    for g in range(G):
        lo, hi = shard_range(n, g, G)
        partials.append(M.msm(cname, np.ascontiguousarray(sc[lo:hi]), np.ascontiguousarray(bases[lo:hi]))[0])
    got = _proj_sum(hip, cname, np.stack(partials))
    single = M.msm(cname, sc, bases)
    exp = refc.msm(sc, bases)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(single)), (cname, G)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(exp)), (cname, G)
    assert refc.is_on_curve(got[0])


def test_msm_sharded_world1_torch(hip):
    """icicle_amd.dist.msm_sharded, the function every rank of bench.py --gpus N runs, at world = 1"""
    import torch
    from icicle_amd import dist as D

    C = pyref.BN254
    refc = ref.RefCurve("bn254")
    rng = np.random.default_rng(61)
    n = 5000
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    dev = torch.device("cuda", 0)
    tb = torch.from_numpy(bases.view(np.int32)).to(dev)
    ts = torch.from_numpy(sc.view(np.int32)).to(dev)
    out = D.msm_sharded("bn254", ts, tb, n, 0, 1, None)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32).reshape(1, -1)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases)))


@pytest.mark.parametrize("cname", CURVES)
@pytest.mark.parametrize("G", [1, 2, 8])
@pytest.mark.parametrize("mode", ["partial_sums", "bucket_exchange"])
def test_msm_multi_device_through_c_abi(hip, cname, G, mode):
    """config.ext {"hip_num_devices": G}: the multi-device MSM behind the unchanged <curve>_msm symbol (shards on
    min(G, physical) devices, logical shards share a device; RCCL all-gather when > 1 physical device). Variant E2
    ("hip_msm_exchange_buckets") exchanges bucket slices instead of final partial sums."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(70 + G)
    n = 6007
    pts = list(cached_points(C, n))
    pts[0] = pyref.INF
    bases = points_to_array(C, pts)
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    exp = refc.to_affine(refc.msm(sc, bases))
    lib.create_config_extension.restype = ctypes.c_void_p
    lib.config_extension_set_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    lib.config_extension_set_bool.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_bool]
    lib.destroy_config_extension.argtypes = [ctypes.c_void_p]
    ext = lib.create_config_extension()
    try:
        lib.config_extension_set_int(ext, b"hip_num_devices", G)
        lib.config_extension_set_bool(ext, b"hip_msm_exchange_buckets", mode == "bucket_exchange")
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        got = M.msm(cname, sc, bases, cfg)  # host operands
        assert np.array_equal(refc.to_affine(got), exp), (cname, G, mode, "host")
        assert refc.is_on_curve(got[0])
        # device-resident operands on the calling device + batch of 2 with shared bases
        from icicle_amd.runtime import DeviceVec

        sc2 = np.vstack([sc, to_words(rand_scalars(rng, n, C.r), 8)])
        d_sc, d_b = DeviceVec.from_host(sc2), DeviceVec.from_host(bases)
        cfg2 = hip.MSMConfig.default()
        cfg2.ext = ext
        cfg2.batch_size = 2
        cfg2.are_points_shared_in_batch = True
        got2 = M.msm(cname, d_sc, d_b, cfg2, msm_size=n)
        exp2 = refc.to_affine(refc.msm(sc2, bases, batch=2, shared=True))
        assert np.array_equal(refc.to_affine(got2), exp2), (cname, G, mode, "device batch")
    finally:
        lib.destroy_config_extension(ext)


def test_rccl_binding_with_size_one_communicator(hip):
    """"hip_force_rccl": the RCCL leg of the in-library multi-device MSM (dlopen of librccl, ncclCommInitAll,
    ncclAllGather on the call's stream, k_proj_sum over the gathered partials) run with a communicator of size 1 --
    everything of the P > 1 exchange that a single-GPU box can execute."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib

    C = pyref.BN254
    refc = ref.RefCurve("bn254")
    rng = np.random.default_rng(123)
    n = 4099
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    ext = lib.create_config_extension()
    try:
        lib.config_extension_set_int(ext, b"hip_num_devices", 3)  # 3 logical shards on the one device
        lib.config_extension_set_bool(ext, b"hip_force_rccl", True)
        from icicle_amd._lib import multi_stats

        for xb in (False, True):
            lib.config_extension_set_bool(ext, b"hip_msm_exchange_buckets", xb)
            cfg = hip.MSMConfig.default()
            cfg.ext = ext
            multi_stats(reset=True)
            got = M.msm("bn254", sc, bases, cfg)
            assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases))), xb
            st = multi_stats()
            if xb:  # the bucket exchange ran as ONE send to the own rank: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd of the real librccl
                assert st["exchange_messages"] == 1 and st["exchanged_bucket_bytes"] > 0, st
    finally:
        lib.destroy_config_extension(ext)


@pytest.mark.parametrize("fname,logn,batch", [("babybear", 12, 1), ("babybear", 17, 2), ("koalabear", 20, 1)])
def test_split_ntt_exchanges_on_the_real_rccl_with_one_device(hip, fname, logn, batch):
    """"hip_force_rccl" with hip_num_devices = 1: the split transform (icicle_amd/csrc/ntt_split.hpp) runs its three all-to-all
    exchanges as grouped ncclSend / ncclRecv to the own rank of a size-1 communicator of the REAL librccl.so -- everything of
    the P > 1 exchange a single-GPU box can execute. memcmp with the reference CPU backend."""
    from icicle_amd import ntt as N
    from icicle_amd._lib import lib, multi_stats

    F = pyref.NTT_FIELDS[fname]
    n = 1 << logn
    rng = np.random.default_rng(logn)
    rf = ref.RefNttField(fname)
    root = N.get_root_of_unity(fname, n)
    rf.init_domain(root)
    N.init_domain(fname, root)
    ext = lib.create_config_extension()
    try:
        lib.config_extension_set_int(ext, b"hip_num_devices", 1)
        lib.config_extension_set_bool(ext, b"hip_force_rccl", True)
        x = rng.integers(0, F.p, size=batch * n, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size = batch
        cfg.ext = ext
        multi_stats(reset=True)
        y = N.ntt(fname, x, N.FORWARD, cfg)
        st = multi_stats()
        assert st["exchange_messages"] == 3 * batch and st["exchanged_bucket_bytes"] == 3 * batch * n * 4, st
        assert np.array_equal(y, rf.ntt(x, n, 0, batch=batch))
        assert np.array_equal(N.ntt(fname, y, N.INVERSE, cfg), x)
    finally:
        lib.destroy_config_extension(ext)
        N.release_domain(fname)
        rf.release_domain()
