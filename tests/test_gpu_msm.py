"""GPU parity: <curve>_msm through the C ABI vs the reference CPU backend (oracle/_ref) and the
pure-Python definition, on identical seeded inputs. Comparison is on AFFINE limbs (bit-exact),
plus the reference's own checks: projective_eq and is_on_curve (icicle/tests/test_curve_api.cpp:77).
Cases follow the reference tests: sizes 2^k - r (test_curve_api.cpp:40), zero points injected and
Montgomery scalars (wrappers/rust/icicle-core/src/msm/tests.rs:17-24,54-59), batch shared/non-shared
(:92-254), skewed 0/1 scalars (:256-304), bitsize sweep (test_curve_api.cpp:82-123)."""
import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import cached_points, from_words, points_to_array, proj_to_affine_py, rand_scalars, to_words

pytestmark = pytest.mark.gpu
CURVES = ["bn254", "bls12_381", "bls12_377", "grumpkin"]


def _check(hip, cname, scalars, bases, refc, **kw):
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    batch = kw.get("batch", 1)
    cfg = hip.MSMConfig.default()
    cfg.batch_size = batch
    cfg.are_points_shared_in_batch = kw.get("shared", True)
    cfg.c = kw.get("c", 0)
    cfg.bitsize = kw.get("bitsize", 0)
    cfg.are_scalars_montgomery_form = kw.get("scalars_mont", False)
    got = M.msm(cname, scalars, bases, cfg)
    sc_ref = scalars
    exp = kw["expected"]() if "expected" in kw else refc.msm(sc_ref, bases, batch=batch, shared=cfg.are_points_shared_in_batch, bitsize=cfg.bitsize,
                                                               scalars_mont=cfg.are_scalars_montgomery_form)
    ga, ea = refc.to_affine(got), refc.to_affine(exp)
    assert np.array_equal(ga, ea), f"affine mismatch {cname} {kw}"
    for b in range(batch):
        assert refc.is_on_curve(got[b])
        assert refc.projective_eq(got[b], exp[b])
        (_, (x, y, z)) = proj_to_affine_py(C, got[b])
        assert not (x == 0 and y == 0 and z == 0), "(0,0,0) is not a valid identity representative"
    return got


@pytest.mark.parametrize("cname", CURVES)
def test_msm_small_vs_python_definition(hip, cname):
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 17, 64):
        pts = cached_points(C, n)
        sc = rand_scalars(rng, n, C.r)
        got = M.msm(cname, to_words(sc, 8), points_to_array(C, pts))
        aff, _ = proj_to_affine_py(C, got[0])
        assert aff == pyref.msm_naive(C, sc, pts), (cname, n)


@pytest.mark.parametrize("cname", CURVES)
@pytest.mark.parametrize("logn", [8, 12, 14])
def test_msm_vs_reference(hip, cname, logn):
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(100 + logn)
    n = (1 << logn) - int(rng.integers(0, 60))
    bases = points_to_array(C, cached_points(C, n))
    scalars = to_words(rand_scalars(rng, n, C.r), 8)
    _check(hip, cname, scalars, bases, refc)


@pytest.mark.parametrize("cname", CURVES)
def test_msm_reference_generator_distribution(hip, cname):
    """bases from projective_t::rand_host_many (period-100 repetition -> equal points meet in buckets)."""
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(5)
    n = 3000
    bases = refc.generate_affine_points(n)
    scalars = to_words(rand_scalars(rng, n, pyref.CURVES[cname].r), 8)
    _check(hip, cname, scalars, bases, refc)
    _check(hip, cname, scalars, bases, refc, c=5)


@pytest.mark.parametrize("cname", CURVES)
def test_msm_edge_cases(hip, cname):
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(11)
    n = 500
    pts = list(cached_points(C, n))
    # zero points injected (msm/tests.rs:17-24), duplicates and exact negations
    pts[3] = pyref.INF
    pts[77] = pyref.INF
    pts[10] = pts[11]
    pts[20] = pyref.ec_neg(C, pts[21])
    bases = points_to_array(C, pts)
    sc = rand_scalars(rng, n, C.r)
    sc[0] = 0
    sc[1] = 1
    sc[2] = C.r - 1
    sc[10] = sc[11] = 5          # same point, same scalar -> doubling inside a bucket
    sc[20] = sc[21] = 9          # P and -P with the same digit -> cancellation inside a bucket
    _check(hip, cname, to_words(sc, 8), bases, refc)
    # all scalars zero / all bases identity -> identity (0:1:0)-like, never (0,0,0)
    z = M.msm(cname, np.zeros((n, 8), dtype=np.uint32), bases)
    aff, (x, y, zz) = proj_to_affine_py(C, z[0])
    assert aff == pyref.INF and zz == 0 and y != 0
    z = M.msm(cname, to_words(sc, 8), np.zeros_like(bases))
    aff, (x, y, zz) = proj_to_affine_py(C, z[0])
    assert aff == pyref.INF and zz == 0 and y != 0
    # all points equal, all scalars equal (every bucket add is a doubling or hits the same x)
    same = points_to_array(C, [pts[5]] * 64)
    _check(hip, cname, to_words([7] * 64, 8), same, refc)
    # size 0
    z = M.msm(cname, np.zeros((0, 8), dtype=np.uint32), np.zeros((0, 2 * C.limbs_q), dtype=np.uint32), msm_size=0)
    aff, (x, y, zz) = proj_to_affine_py(C, z[0])
    assert aff == pyref.INF and y != 0


@pytest.mark.parametrize("cname", CURVES)
def test_msm_skewed_and_bitsize(hip, cname):
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(13)
    n = 4000
    bases = points_to_array(C, cached_points(C, n))
    # skewed: mostly 0/1 scalars with bitsize = 1 (msm/tests.rs:256-304)
    sc = [int(v) for v in rng.integers(0, 2, size=n)]
    _check(hip, cname, to_words(sc, 8), bases, refc, bitsize=1)
    _check(hip, cname, to_words(sc, 8), bases, refc)
    for bits in (2, 7, 31, 32, 33, 64, 129, 200, C.r.bit_length() - 1):
        sc = rand_scalars(rng, 300, C.r, bits=bits)
        _check(hip, cname, to_words(sc, 8), bases[:300], refc, bitsize=bits)
    # scalars WIDER than bitsize: only the low bitsize bits count (cpu_msm.hpp:288; test_curve_api.cpp:82-123 feeds
    # full-width scalars to every bitsize from 1 to NBITS)
    full = to_words(rand_scalars(rng, 300, C.r), 8)
    for bits in (1, 2, 5, 19, 20, 21, 31, 32, 33, 63, 64, 65, 127, 200, C.r.bit_length() - 1, C.r.bit_length()):
        _check(hip, cname, full, bases[:300], refc, bitsize=bits)


@pytest.mark.parametrize("cname", CURVES)
def test_msm_batch_and_montgomery(hip, cname):
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(17)
    n, batch = 700, 3
    pts = cached_points(C, n * batch)
    sc = to_words(rand_scalars(rng, n * batch, C.r), 8)
    _check(hip, cname, sc, points_to_array(C, pts[:n]), refc, batch=batch, shared=True)
    _check(hip, cname, sc, points_to_array(C, pts), refc, batch=batch, shared=False)
    # scalars in Montgomery form (R = 2^256), as the Rust test always does
    scm = refc.scalars_to_montgomery(sc[:n])
    got = _check(hip, cname, scm, points_to_array(C, pts[:n]), refc, scalars_mont=True)
    plain = _check(hip, cname, sc[:n], points_to_array(C, pts[:n]), refc)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(plain))


@pytest.mark.parametrize("cname", CURVES)
def test_msm_explicit_c_sweep(hip, cname):
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(19)
    n = 1000
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    from icicle_amd import msm as M

    base = None
    for c in (2, 3, 7, 10, 13, 16):
        cfg = hip.MSMConfig.default()
        cfg.c = c
        got = refc.to_affine(M.msm(cname, sc, bases, cfg))
        if base is None:
            base = refc.to_affine(refc.msm(sc, bases))
        assert np.array_equal(got, base), c


@pytest.mark.parametrize("cname", CURVES)
def test_msm_device_resident_async(hip, cname):
    """scalars, bases and results on device, is_async on a created stream (msm/tests.rs:60-75)."""
    from icicle_amd import msm as M
    from icicle_amd.runtime import DeviceVec, Stream

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(23)
    n = 2048
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    d_sc, d_b = DeviceVec.from_host(sc), DeviceVec.from_host(bases)
    d_res = DeviceVec(3 * C.limbs_q * 4)
    st = Stream()
    cfg = hip.MSMConfig.default()
    cfg.stream = st.handle
    cfg.is_async = True
    M.msm(cname, d_sc, d_b, cfg, results=d_res, msm_size=n)
    st.synchronize()
    got = d_res.to_host(shape=(1, 3 * C.limbs_q))
    exp = refc.msm(sc, bases)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(exp))
    st.destroy()


@pytest.mark.parametrize("cname", CURVES)
def test_msm_precompute(hip, cname):
    """msm_precompute_bases + precompute_factor (test_curve_api.cpp:126-171): self-consistency on
    our device plus equality with the reference result computed WITHOUT precompute."""
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(29)
    n = 600
    pts = list(cached_points(C, n))
    pts[4] = pyref.INF
    bases = points_to_array(C, pts)
    sc = to_words(rand_scalars(rng, n * 2, C.r), 8)
    exp = refc.to_affine(refc.msm(sc, bases, batch=2, shared=True))
    for pf in (2, 3, 8) + ((40,) if cname == "bn254" else ()):  # (40: more outputs per base than one shared inversion holds)
        cfg = hip.MSMConfig.default()
        cfg.precompute_factor = pf
        cfg.batch_size = 2
        pre = M.precompute_bases(cname, bases, cfg)
        assert pre.shape == (n * pf, 2 * C.limbs_q)
        assert np.array_equal(pre[::pf], bases)  # j = 0 entry is the point itself
        got = M.msm(cname, sc, pre, cfg)
        assert np.array_equal(refc.to_affine(got), exp), pf


@pytest.mark.parametrize("cname", CURVES)
def test_msm_precompute_non_shared_batch(hip, cname):
    """per-MSM base tables as the Rust suite builds them (wrappers/rust/icicle-core/src/msm/tests.rs:195-220):
    msm_precompute_bases over batch * n bases with batch_size / are_points_shared_in_batch = false set on BOTH calls and
    nof_bases = the bases of ONE MSM (msm/mod.rs:296); and a different msm_size on the same table with config.c pinned"""
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(33)
    n, batch, pf = 300, 3, 4
    bases = points_to_array(C, cached_points(C, n * batch))
    sc = to_words(rand_scalars(rng, n * batch, C.r), 8)
    exp = refc.to_affine(refc.msm(sc, bases, batch=batch, shared=False))
    cfg = hip.MSMConfig.default()
    cfg.precompute_factor = pf
    cfg.batch_size = batch
    cfg.are_points_shared_in_batch = False
    pre = M.precompute_bases(cname, bases, cfg)  # nof_bases = n, batch * n bases extended
    assert pre.shape[0] == n * batch * pf and pre[-1].any()
    got = M.msm(cname, sc, pre, cfg)
    assert np.array_equal(refc.to_affine(got), exp)
    # the first 100 bases of a table built with an explicit c, as a single MSM of a different size
    cfg.c = 9
    pre = M.precompute_bases(cname, bases, cfg)
    cfg1 = hip.MSMConfig.default()
    cfg1.precompute_factor = pf
    cfg1.c = 9
    got1 = M.msm(cname, np.ascontiguousarray(sc[:100]), np.ascontiguousarray(pre[: 100 * pf]), cfg1)
    assert np.array_equal(refc.to_affine(got1), refc.to_affine(refc.msm(np.ascontiguousarray(sc[:100]), np.ascontiguousarray(bases[:100]))))


def test_table_overwritten_through_the_runtime_api_is_not_looked_up_with_its_old_window_size(hip):
    """msm() with precompute_factor > 1 and c = 0 takes the window size of a table this process wrote from the table
    registry. A device buffer that held a table built with c = 5 and is then overwritten (icicle_copy_to_device) with a table built
    with the default c must not be run with c = 5: writes through the runtime API drop the entries they overlap."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check
    from icicle_amd.runtime import DeviceVec

    C = pyref.BN254
    refc = ref.RefCurve("bn254")
    rng = np.random.default_rng(515)
    n, pf = 700, 2
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    exp = refc.to_affine(refc.msm(sc, bases))
    cfg5 = hip.MSMConfig.default()
    cfg5.precompute_factor, cfg5.c = pf, 5
    d_tab = DeviceVec(bases.nbytes * pf)
    d_in = DeviceVec.from_host(bases)
    try:
        M.precompute_bases("bn254", d_in, cfg5, output=d_tab, nof_bases=n)  # registered: this range holds a c = 5 table
        cfg0 = hip.MSMConfig.default()
        cfg0.precompute_factor = pf
        host_tab = M.precompute_bases("bn254", bases, cfg0)  # a table with the default window size, built on the host side
        check(lib.icicle_copy_to_device(d_tab.ptr, host_tab.ctypes.data, host_tab.nbytes))
        got = M.msm("bn254", sc, d_tab, cfg0, msm_size=n)  # c = 0: must be derived from msm_size, not read from the stale entry
        assert np.array_equal(refc.to_affine(got), exp)
    finally:
        d_tab.free(), d_in.free()


def test_msm_precompute_refuses_a_total_count_that_overruns_the_device_buffers(hip):
    """msm_precompute_bases takes nof_bases as the bases of ONE MSM (both wrappers' convention); a caller
    that hands the TOTAL with per-MSM bases would be read / written batch_size times past its buffers. With device buffers
    the overrun is provable (hipMemGetAddressRange) and the call returns INVALID_ARGUMENT instead of corrupting memory."""
    from icicle_amd import msm as M
    from icicle_amd.runtime import DeviceVec

    C = pyref.BN254
    n, batch, pf = 256, 4, 2
    bases = points_to_array(C, cached_points(C, n * batch))
    d_in = DeviceVec.from_host(bases)
    d_out = DeviceVec(bases.nbytes * pf)
    cfg = hip.MSMConfig.default()
    cfg.precompute_factor, cfg.batch_size, cfg.are_points_shared_in_batch = pf, batch, False
    try:
        M.precompute_bases("bn254", d_in, cfg, output=d_out, nof_bases=n)  # the wrappers' convention: fits exactly
        with pytest.raises(RuntimeError):
            M.precompute_bases("bn254", d_in, cfg, output=d_out, nof_bases=n * batch)  # total count: 4 x past both buffers
        got = d_out.to_host(shape=(n * batch * pf, 16))
        exp = M.precompute_bases("bn254", bases, cfg)
        assert np.array_equal(got, exp)  # the refused call wrote nothing
    finally:
        d_in.free()
        d_out.free()


@pytest.mark.parametrize("cname", CURVES)
def test_msm_on_a_table_of_another_size_finds_its_window(hip, cname):
    """with precompute_factor > 1 and config.c = 0 both calls derive c from their OWN size, so an MSM over a prefix
    of a table, or a batched MSM on a table precomputed with batch_size 1, used a shift the table was not built with --
    silently wrong. msm_precompute_bases now records (address range -> c) and msm() looks its bases pointer up first."""
    from icicle_amd import msm as M
    import ctypes

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(35)
    n, pf = 1 << 14, 8
    bases = points_to_array(C, cached_points(C, 2048))
    bases = np.ascontiguousarray(np.tile(bases, (n // 2048, 1)))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    import torch

    L2 = 2 * M.LIMBS[cname]
    dev = torch.device("cuda", 0)
    cfg = hip.MSMConfig.default()
    cfg.precompute_factor = pf
    pre_h = M.precompute_bases(cname, bases, cfg)  # a HOST table, c chosen for 2^14
    pre_d = torch.empty((n * pf, L2), dtype=torch.int32, device=dev)  # the same table written to DEVICE memory
    M.precompute_bases(cname, bases, cfg, output=pre_d.data_ptr(), nof_bases=n)
    assert np.array_equal(pre_d.cpu().numpy().view(np.uint32), pre_h)
    c_big, c_small, nw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    hip.lib.icicle_hip_msm_plan(n, C.r.bit_length(), ctypes.byref(cfg), ctypes.byref(c_big), ctypes.byref(nw))
    hip.lib.icicle_hip_msm_plan(100, C.r.bit_length(), ctypes.byref(cfg), ctypes.byref(c_small), ctypes.byref(nw))
    assert c_big.value != c_small.value  # (otherwise this test proves nothing)
    exp100 = refc.to_affine(refc.msm(np.ascontiguousarray(sc[:100]), np.ascontiguousarray(bases[:100])))
    # (1) a prefix of the table as an MSM of 100 terms, c left to the backend: host table (recognised at its start) and device table
    assert np.array_equal(refc.to_affine(M.msm(cname, np.ascontiguousarray(sc[:100]), pre_h[: 100 * pf], cfg)), exp100)
    assert np.array_equal(refc.to_affine(M.msm(cname, np.ascontiguousarray(sc[:100]), pre_d.data_ptr(), cfg, msm_size=100)), exp100)
    # (2) an interior slice of the DEVICE table (starts at base 512)
    got = M.msm(cname, np.ascontiguousarray(sc[512:812]), pre_d.data_ptr() + 512 * pf * L2 * 4, cfg, msm_size=300)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(np.ascontiguousarray(sc[512:812]), np.ascontiguousarray(bases[512:812]))))
    # (3) the table precomputed in ONE call with batch_size 1, used as per-MSM tables of a batch of 16 MSMs of 2^10
    cfgb = hip.MSMConfig.default()
    cfgb.precompute_factor = pf
    cfgb.batch_size = 16
    cfgb.are_points_shared_in_batch = False
    got = M.msm(cname, sc, pre_d.data_ptr(), cfgb, msm_size=n // 16)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases, batch=16, shared=False)))
    got = M.msm(cname, sc, pre_h, cfgb)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases, batch=16, shared=False)))


def test_generated_points_are_distinct_multiples_of_g(hip):
    from icicle_amd import msm as M

    C = pyref.BN254
    pts = M.generate_affine_points("bn254", 100, k0=5)
    exp = points_to_array(C, pyref.gen_points(C, 100, k0=5))
    assert np.array_equal(pts, exp)


def _full_size_inputs(dev, cname="bn254", logn=26, top=0x30644E72):
    import torch
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check

    L = M.LIMBS[cname]
    n = 1 << logn
    bases = torch.empty((n, 2 * L), dtype=torch.int32, device=dev)
    check(getattr(lib, f"{cname}_hip_generate_affine_points")(bases.data_ptr(), n, 12345, True, None))
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    sc = torch.randint(-(2 ** 31), 2 ** 31, (n, 8), dtype=torch.int32, device=dev, generator=g)
    sc[:, 7] = torch.randint(0, top, (n,), dtype=torch.int32, device=dev, generator=g)
    torch.cuda.synchronize()
    return sc, bases


def _job_full_size(pool, hip, dev):
    """the reference CPU backend on the FULL 2^26 inputs (BASELINE configs[1] says "bit-exact vs CPU": compared, not inferred);
    a background job of tests/refpool.py"""
    sc, bases = _full_size_inputs(dev)
    pool.submit_msm("bn254_26_uniform", "bn254", np.ascontiguousarray(sc.cpu().numpy().view(np.uint32)), np.ascontiguousarray(bases.cpu().numpy().view(np.uint32)))


REF_JOBS = {"bn254_26_uniform": (4, _job_full_size)}


@pytest.mark.refjob("bn254_26_uniform", order=15)
@pytest.mark.parametrize("cname,logn,top", [("bn254", 26, 0x30644E72)])  # (config 3 runs whole: test_gpu_fullsize_configs.py)
def test_msm_full_size_split_property(hip, refpool, cname, logn, top):
    """BASELINE config 1 size (2^26 BN254), inputs resident in HBM: the size-independent property MSM(all) == MSM(first half) +
    MSM(second half), with the halves combined by the reference's own ecadd, AND the reference CPU backend run on the full inputs."""
    import ctypes
    import torch
    from icicle_amd import msm as M

    refc = ref.RefCurve(cname)
    L = M.LIMBS[cname]
    n = 1 << logn
    dev = torch.device("cuda", 0)
    sc, bases = _full_size_inputs(dev, cname, logn, top)

    def run(lo, hi):
        cfg = hip.MSMConfig.default()
        out = np.zeros((1, 3 * L), dtype=np.uint32)
        M.msm(cname, sc[lo:hi].data_ptr(), bases[lo:hi].data_ptr(), cfg, results=out, msm_size=hi - lo)
        return out

    full, a, b = run(0, n), run(0, n // 2), run(n // 2, n)
    s = np.zeros(3 * L, dtype=np.uint32)
    getattr(refc.lib, f"{cname}_ecadd")(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(s.ctypes.data))
    assert np.array_equal(refc.to_affine(full), refc.to_affine(s.reshape(1, 3 * L)))
    assert refc.is_on_curve(full[0])
    del sc, bases
    exp = refpool.result("bn254_26_uniform")  # ... and the byte compare itself: the reference on the full inputs
    assert np.array_equal(refc.to_affine(full), refc.to_affine(exp)), f"{cname} 2^{logn}: GPU result differs from the reference CPU backend"
    assert refc.projective_eq(full[0], exp[0])


@pytest.mark.parametrize("cname", CURVES)
def test_msm_skewed_large_overflow_segments(hip, cname):
    """2^18 scalars, almost all equal to 1 or 2 (msm/tests.rs:256-304 style skew): two buckets hold ~2^17
    points each, far beyond the per-thread segment, so the overflow-segment path carries the result."""
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(31)
    n = 1 << 18
    bases = M.generate_affine_points(cname, n, k0=999)
    vals = rng.integers(1, 3, size=n)
    sc = np.zeros((n, 8), dtype=np.uint32)
    sc[:, 0] = vals
    sc[::1000] = to_words(rand_scalars(rng, len(sc[::1000]), C.r), 8)
    _check(hip, cname, sc, bases, refc)
    _check(hip, cname, sc, bases, refc, c=16)


def test_concurrent_host_threads(hip):
    """four host threads issue MSMs and NTTs at the same time (ctypes releases the GIL during the call): every
    call leases its own temporaries, so the results must be those of the serial runs"""
    import threading

    from icicle_amd import msm as M
    from icicle_amd import ntt as N
    from icicle_amd import runtime

    C = pyref.BN254
    rng = np.random.default_rng(91)
    n = 3000
    bases = points_to_array(C, cached_points(C, n))
    scal = [to_words(rand_scalars(rng, n, C.r), 8) for _ in range(4)]
    serial = [M.msm("bn254", s, bases) for s in scal]
    F = pyref.BABYBEAR
    N.init_domain("babybear", N.get_root_of_unity("babybear", 1 << 14))
    xs = [rng.integers(0, F.p, size=1 << 14, dtype=np.uint32) for _ in range(4)]
    serial_ntt = [N.ntt("babybear", x, N.FORWARD) for x in xs]
    out, out_ntt, errs = [None] * 4, [None] * 4, []

    def work(i):
        try:
            runtime.set_device(0)  # the active device is per host thread (icicle_set_device)
            for _ in range(5):
                out[i] = M.msm("bn254", scal[i], bases)
                out_ntt[i] = N.ntt("babybear", xs[i], N.FORWARD)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    N.release_domain("babybear")
    assert not errs, errs
    refc = ref.RefCurve("bn254")
    for i in range(4):
        assert np.array_equal(refc.to_affine(out[i]), refc.to_affine(serial[i]))
        assert np.array_equal(out_ntt[i], serial_ntt[i])


def _large_batch_inputs(cname):
    """600 MSMs of 2^10 terms on shared bases, 520 of them on per-MSM bases (host arrays; the bases come from the GPU generator)"""
    from icicle_amd import msm as M

    rng = np.random.default_rng(37)
    n, batch, nb2 = 1 << 10, 600, 520
    bases = M.generate_affine_points(cname, n, k0=4242)
    words = rng.integers(0, 1 << 32, size=(n * batch, 8), dtype=np.uint64).astype(np.uint32)
    words[:, 7] &= 0x0FFFFFFF  # < r for both curves
    words[::97] = 0
    bases2 = M.generate_affine_points(cname, n * nb2, k0=99)
    return n, batch, nb2, bases, words, bases2


def _job_large_batch(cname):
    def start(pool, hip, dev):  # the two reference batch MSMs (~15-40 s of host time per curve): background jobs of tests/refpool.py
        n, batch, nb2, bases, words, bases2 = _large_batch_inputs(cname)
        pool.submit_msm(f"large_batch_{cname}_shared", cname, words, bases, lane="msm_small", batch=batch, shared=True)
        pool.submit_msm(f"large_batch_{cname}_per_msm", cname, np.ascontiguousarray(words[: n * nb2]), bases2, lane="msm_small", batch=nb2, shared=False)
    return start


for _c in CURVES:
    REF_JOBS[f"large_batch_{_c}_shared"] = (8, _job_large_batch(_c))
    REF_JOBS[f"large_batch_{_c}_per_msm"] = (8, REF_JOBS[f"large_batch_{_c}_shared"][1])


@pytest.mark.refjob(*[f"large_batch_{c}_{k}" for c in CURVES for k in ("shared", "per_msm")], order=8)
@pytest.mark.parametrize("cname", CURVES)
def test_msm_large_batch_fused_digit_histogram(hip, refpool, cname):
    """batch >= 512 with enough scalar chunks takes the fused k_digits_count path (digits + pass-A histogram in one
    kernel; otherwise only reached by the 2^26 run): 600 MSMs of 2^10 terms, shared and per-MSM bases, every result
    against the reference CPU backend (the perf matrix benchmarks this shape: 1024 x 2^12)."""
    refc = ref.RefCurve(cname)
    n, batch, nb2, bases, words, bases2 = _large_batch_inputs(cname)
    _check(hip, cname, words, bases, refc, batch=batch, shared=True, expected=lambda: refpool.result(f"large_batch_{cname}_shared"))
    _check(hip, cname, np.ascontiguousarray(words[: n * nb2]), bases2, refc, batch=nb2, shared=False, expected=lambda: refpool.result(f"large_batch_{cname}_per_msm"))


@pytest.mark.parametrize("curve_id", [0, 1])
def test_inplace_asm_products_with_aliased_and_constant_operands(hip, curve_id):
    """the in-place asm products overwrite their read-write operand while the other operands are still read;
    with early-clobber operands an aliased input (a <- a * a) or a constant one must still give the out-of-place value.
    4096 pseudo-random operand sets per curve, checked on the device."""
    import ctypes
    from icicle_amd._lib import lib, check

    bad = ctypes.c_int(-1)
    check(lib.icicle_hip_selftest_inplace_products(curve_id, ctypes.byref(bad)), "selftest")
    assert bad.value == 0


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_msm_window_size_22_forced(hip, cname):
    """config.c = 22 (12 windows of a 254 / 255-bit scalar; pass B of the sort then ranks into 2^11 bins, two per thread):
    never chosen by the plan -- it measured slower, profiles/r04_msm_csweep.txt -- but a caller may ask for it. 2^18 uniform
    scalars plus the skewed mix (two hot buckets -> overflow segments) against the reference CPU backend."""
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(2222)
    n = (1 << 18) - 7
    bases = M.generate_affine_points(cname, n, k0=31337)
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    _check(hip, cname, sc, bases, refc, c=22)
    sc[: n // 2, 1:] = 0
    sc[: n // 2, 0] = rng.integers(1, 3, size=n // 2)
    _check(hip, cname, sc, bases, refc, c=22)


@pytest.mark.parametrize("cname", CURVES)
@pytest.mark.parametrize("nwin", [12, 13, 17, 23, 31, 40])
def test_msm_mixed_window_widths(hip, cname, nwin):
    """Round 5: window widths that add up to the scalar bits exactly (the top x windows one bit wider) with the reference's
    'negate scalar and point when the top bit is set' trick (cpu_msm.hpp:276-277). The cost model picks such a plan from
    ~2^25 terms up (BN254 2^26: 10 x 21 + 2 x 22 bits); MSMConfig.ext "hip_msm_windows" forces one at a size the oracle
    finishes in seconds. Scalars include r - 1, 2^253 +- 1, the all-ones digits and 0 / 1; bases include the identity."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(1000 + nwin)
    n = 5000
    pts = list(cached_points(C, n))
    pts[3] = pyref.INF
    bases = points_to_array(C, pts)
    vals = rand_scalars(rng, n, C.r)
    top = C.r.bit_length() - 1
    edge = [C.r - 1, C.r - 2, (1 << top), (1 << top) - 1, (1 << top) + 1, 0, 1, 2, (1 << (top - 1)), C.r // 2, C.r // 2 + 1, ((1 << top) - 1) // 3]
    for k, v in enumerate(edge):
        vals[10 + k] = v % C.r
    sc = to_words(vals, 8)
    ext = lib.create_config_extension()
    lib.config_extension_set_int(ext, b"hip_msm_windows", nwin)
    try:
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        got = M.msm(cname, sc, bases, cfg)
    finally:
        lib.destroy_config_extension(ext)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases)))
    assert refc.is_on_curve(got[0])


@pytest.mark.parametrize("cname,g2", [(c, False) for c in CURVES] + [("bn254", True), ("bls12_381", True)])
def test_window_combine_on_the_gpu_equals_the_host_side_combine(hip, cname, g2):
    """a single MSM whose result goes to the host -- or stays on the device of a SYNCHRONOUS call -- combines its window
    sums on one host core (the host build of ec.hpp: dbl_jac + add), an asynchronous device-resident result keeps k_final on the
    GPU. Both routes on the same inputs, for every curve and G2, with a base table (precompute_factor 4) and with a forced
    mixed-width plan; the host route is also checked against the reference."""
    import torch
    from icicle_amd import msm as M
    from icicle_amd._lib import lib

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname, g2=g2)
    rng = np.random.default_rng(909)
    n = 3001
    bases = M.generate_affine_points(cname, n, k0=4711, g2=g2)
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    L = bases.shape[1] // 2
    dev = torch.device("cuda", 0)
    d_sc = torch.from_numpy(sc.view(np.int32)).to(dev)

    def both(table, **cfgkw):
        d_b = torch.from_numpy(table.view(np.int32)).to(dev)
        outs = []
        for on_device, is_async in ((False, False), (True, True), (True, False)):
            cfg = hip.MSMConfig.default()
            for k, v in cfgkw.items():
                setattr(cfg, k, v)
            cfg.is_async = is_async
            if on_device:
                d_out = torch.zeros(3 * L, dtype=torch.int32, device=dev)
                M.msm(cname, d_sc.data_ptr(), d_b.data_ptr(), cfg, results=d_out.data_ptr(), msm_size=n, g2=g2)
                torch.cuda.synchronize()
                outs.append(d_out.cpu().numpy().view(np.uint32).reshape(1, -1))
            else:
                out = np.zeros((1, 3 * L), dtype=np.uint32)
                M.msm(cname, d_sc.data_ptr(), d_b.data_ptr(), cfg, results=out, msm_size=n, g2=g2)
                outs.append(out)
        aff = [refc.to_affine(o) for o in outs]
        assert np.array_equal(aff[0], aff[1]), (cname, g2, cfgkw, "host combine differs from k_final")
        assert np.array_equal(aff[0], aff[2]), (cname, g2, cfgkw, "synchronous device-resident result")
        return aff[0]

    exp = refc.to_affine(refc.msm(sc, bases))
    assert np.array_equal(both(bases), exp)
    cfgp = hip.MSMConfig.default()
    cfgp.precompute_factor = 4
    table = M.precompute_bases(cname, bases, cfgp, g2=g2)
    assert np.array_equal(both(table, precompute_factor=4), exp)
    if not g2:
        ext = lib.create_config_extension()
        lib.config_extension_set_int(ext, b"hip_msm_windows", 19)
        try:
            assert np.array_equal(both(bases, ext=ext), exp)
        finally:
            lib.destroy_config_extension(ext)


@pytest.mark.parametrize("cname,g2", [(c, False) for c in CURVES] + [("bn254", True), ("bls12_381", True), ("bls12_377", True)])
def test_every_size_and_precompute_factor_the_reference_tests_can_draw(hip, cname, g2):
    """icicle/tests/test_curve_api.cpp draws ONE point of a small space per run (clock seed): MSM_test N = 2^12 - rand(0..60), batch 1
    (:36-79); MSM_PRE_COMPUTE_test the same N with batch 3 on shared bases and precompute_factor = rand(1..8) (:125-170); bases from
    projective_t::rand_host_many (100 points repeated). Here: every N for the plain MSM, every precompute_factor on eight of the sizes
    (both ends, the size classes around a multiple of 64 / 100) for the table path -- against the reference CPU backend."""
    from icicle_amd import msm as M

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname, g2=g2)
    rng = np.random.default_rng(4096)
    nmax = 1 << 12
    bases_all = refc.generate_affine_points(nmax)
    sc_all = to_words(rand_scalars(rng, 3 * nmax, C.r), 8)
    bad = []
    for r in range(0, 61):
        n = nmax - r
        got = M.msm(cname, np.ascontiguousarray(sc_all[:n]), np.ascontiguousarray(bases_all[:n]), g2=g2)
        exp = refc.msm(np.ascontiguousarray(sc_all[:n]), np.ascontiguousarray(bases_all[:n]))
        if not (np.array_equal(refc.to_affine(got), refc.to_affine(exp)) and refc.is_on_curve(got[0])):
            bad.append(("msm", n))
    for r in (0, 1, 4, 32, 33, 59, 60, 17):
        n = nmax - r
        b = np.ascontiguousarray(bases_all[:n])
        sc = np.ascontiguousarray(sc_all[: 3 * n])
        exp = refc.to_affine(refc.msm(sc, b, batch=3, shared=True))
        for pf in range(1, 9):
            cfg = hip.MSMConfig.default()
            cfg.batch_size, cfg.are_points_shared_in_batch, cfg.precompute_factor = 3, True, pf
            table = M.precompute_bases(cname, b, cfg, g2=g2)
            got = M.msm(cname, sc, table, cfg, g2=g2)
            if not np.array_equal(refc.to_affine(got), exp):
                bad.append(("precompute", n, pf))
    assert not bad, bad[:20]


def test_idle_workspace_decays(hip):
    """the temporaries of a large call stayed cached until an allocation failed or the caller asked. Arenas idle
    for ICICLE_HIP_WORKSPACE_DECAY_S seconds (default 30) are now given back by the next call that leases a temporary."""
    import os
    import subprocess
    import sys

    code = r"""
import ctypes, time, numpy as np
import icicle_amd
from icicle_amd import msm as M, ntt as N, runtime
from icicle_amd._lib import lib, check
runtime.set_device(0)
n = 1 << 18
bases = M.generate_affine_points("bn254", n, k0=3)
sc = np.random.default_rng(1).integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32); sc[:, 7] &= 0x0FFFFFFF
M.msm("bn254", sc, bases)
c = ctypes.c_size_t()
check(lib.icicle_hip_workspace_bytes(ctypes.byref(c))); big = c.value
time.sleep(1.6)
x = np.arange(16, dtype=np.uint32)
N.init_domain("babybear", N.get_root_of_unity("babybear", 16)); N.ntt("babybear", x, N.FORWARD)   # any call that leases a temporary
check(lib.icicle_hip_workspace_bytes(ctypes.byref(c))); small = c.value
print("DECAY", big, small)
assert big > (32 << 20) and small < big // 4, (big, small)
# ... and without any further msm() / ntt() at all (ADVICE r05: a process that moves on to non-icicle work): asking for the free
# memory, or a plain icicle_malloc, gives idle arenas back as well
M.msm("bn254", sc, bases)
check(lib.icicle_hip_workspace_bytes(ctypes.byref(c))); big2 = c.value
time.sleep(1.6)
tot, free = ctypes.c_size_t(), ctypes.c_size_t()
check(lib.icicle_get_available_memory(ctypes.byref(tot), ctypes.byref(free)))
check(lib.icicle_hip_workspace_bytes(ctypes.byref(c))); after = c.value
print("DECAY2", big2, after)
assert big2 > (32 << 20) and after < big2 // 4, (big2, after)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ICICLE_HIP_WORKSPACE_DECAY_S="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0 and "DECAY" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("quad", ["1", "0"])
def test_bucket_reduction_routes_across_the_cost_model_boundary(hip, quad):
    """Round 6: single MSMs of a uniform plan reduce their buckets with four lanes per bucket column (k_reduce_wave_quad /
    k_reduce_window_quad, msm_impl.hpp) while the reduction is a latency chain -- the wave kernel up to ~2^18 terms by a cost model, the
    window kernel whenever a window has at most 64 chunks -- and with the one-lane kernels beyond. Sizes on both sides of every
    switch (window sizes 8, 15, 16, 17; free chunk sizes: 57 chunks of 288 buckets at 2^16), two curves, against the reference CPU
    backend; `ICICLE_HIP_MSM_REDUCE_QUAD` is read once per process, hence a child per setting: "0" keeps the one-lane route of
    rounds 1-5 covered at the small sizes where the default no longer takes it."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import icicle_amd as hip
from icicle_amd import msm as M, runtime
from oracle import pyref, ref
from tests.util import cached_points, points_to_array, rand_scalars, to_words
runtime.set_device(0)
for cname, logns in (("bn254", (3, 6, 10, 12, 13, 15, 16, 17, 18, 19, 20)), ("bls12_381", (9, 13, 16, 18))):
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(606)
    for logn in logns:
        n = (1 << logn) - int(rng.integers(0, 5))
        period = min(n, 4096)
        pts = points_to_array(C, cached_points(C, period))
        bases = np.ascontiguousarray(np.tile(pts, ((n + period - 1) // period, 1))[:n])
        sc = to_words(rand_scalars(rng, n, C.r), 8)
        got = M.msm(cname, sc, bases)
        exp = refc.msm(sc, bases)
        assert np.array_equal(refc.to_affine(got), refc.to_affine(exp)), (cname, logn)
print("REDUCE OK")
""" % root
    env = dict(os.environ)
    env["ICICLE_HIP_MSM_REDUCE_QUAD"] = quad
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0 and "REDUCE OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
