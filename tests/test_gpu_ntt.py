"""GPU parity: <field>_ntt through the C ABI vs the reference CPU backend, memcmp-exact
(icicle/tests/test_mod_arithmetic_api.h:614-695 draws logn, batch, columns_batch, in-place, dir,
ordering and coset at random and memcmp's main vs reference; same matrix here, seeded)."""
import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu
FIELDS = ["babybear", "koalabear"]
DOMAIN_LOG = {"babybear": 20, "koalabear": 20}


@pytest.fixture(scope="module", params=FIELDS)
def env(request, hip):
    from icicle_amd import ntt as N

    fname = request.param
    F = pyref.NTT_FIELDS[fname]
    rf = ref.RefNttField(fname)
    root = N.get_root_of_unity(fname, 1 << DOMAIN_LOG[fname])
    assert root == rf.get_root_of_unity(1 << DOMAIN_LOG[fname]) == pyref.omega(F, DOMAIN_LOG[fname])
    N.init_domain(fname, root)
    N.init_domain(fname, root)  # second init is a silent success (cpu_ntt_domain.h:69)
    rf.init_domain(root)
    yield fname, F, rf, N
    N.release_domain(fname)
    rf.release_domain()


def test_rou_from_domain(env):
    fname, F, rf, N = env
    for logn in (0, 1, 5, DOMAIN_LOG[fname]):
        assert N.get_root_of_unity_from_domain(fname, logn) == rf.get_root_of_unity_from_domain(logn)


def test_ntt_vs_python_definition(env, hip):
    fname, F, rf, N = env
    rng = np.random.default_rng(1)
    for logn in (0, 1, 2, 5, 8):
        n = 1 << logn
        x = rng.integers(0, F.p, size=n, dtype=np.uint32)
        y = N.ntt(fname, x, N.FORWARD)
        assert [int(v) for v in y] == pyref.ntt_naive(F, [int(v) for v in x], pyref.omega(F, logn))
        back = N.ntt(fname, y, N.INVERSE)
        assert np.array_equal(back, x)


@pytest.mark.parametrize("logn", [0, 1, 3, 6, 10, 12, 13, 15, 17])
def test_ntt_matrix_vs_reference(env, hip, logn):
    fname, F, rf, N = env
    rng = np.random.default_rng(1000 + logn)
    n = 1 << logn
    for trial in range(6):
        batch = int(rng.choice([1, 2, 4, 7]))
        columns = bool(rng.integers(0, 2))
        ordering = int(rng.integers(0, 6))
        direction = int(rng.integers(0, 2))
        coset = 1 if rng.integers(0, 2) else int(rng.integers(2, F.p))
        x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.columns_batch, cfg.ordering, cfg.coset_gen = batch, columns, ordering, coset
        got = N.ntt(fname, x, direction, cfg)
        exp = rf.ntt(x, n, direction, batch=batch, columns_batch=columns, ordering=ordering, coset_gen=coset)
        assert np.array_equal(got, exp), (fname, logn, batch, columns, ordering, direction, coset)


@pytest.mark.parametrize("logn", [4, 11, 14])
def test_ntt_device_inplace_async(env, hip, logn):
    fname, F, rf, N = env
    from icicle_amd.runtime import DeviceVec, Stream

    rng = np.random.default_rng(77 + logn)
    n, batch = 1 << logn, 3
    x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
    d = DeviceVec.from_host(x)
    st = Stream()
    cfg = hip.NTTConfigU32.default()
    cfg.batch_size, cfg.stream, cfg.is_async = batch, st.handle, True
    N.ntt(fname, d, N.FORWARD, cfg, out=d, size=n)  # in place on device
    st.synchronize()
    assert np.array_equal(d.to_host(), rf.ntt(x, n, 0, batch=batch))
    cfg.ordering = N.kRN
    cfg.coset_gen = 3
    N.ntt(fname, d, N.INVERSE, cfg, out=d, size=n)
    st.synchronize()
    assert np.array_equal(d.to_host(), rf.ntt(rf.ntt(x, n, 0, batch=batch), n, 1, batch=batch, ordering=2, coset_gen=3))
    st.destroy()


def test_ntt_16_byte_lane_passes_and_unaligned_fallback(env, hip):
    """2^16 = two 2^8 passes with 32-column tiles: on 16-byte-aligned device buffers k_ntt_fast runs its uint4 load /
    store paths (V4), on buffers that start 4 bytes off a 16-byte boundary the 4-byte-lane passes. Both must equal
    the reference CPU backend word for word (test_mod_arithmetic_api.h:694 memcmp)."""
    import torch

    fname, F, rf, N = env
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4242)
    n, batch = 1 << 16, 5
    x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
    cfg = hip.NTTConfigU32.default()
    cfg.batch_size = batch
    for direction, ordering in ((N.FORWARD, N.kNN), (N.INVERSE, N.kNN), (N.FORWARD, N.kNR)):
        cfg.ordering = ordering
        exp = rf.ntt(x, n, 0 if direction == N.FORWARD else 1, batch=batch, ordering=ordering)
        for shift in (0, 1):  # element offset of both buffers inside 16-byte-aligned allocations
            src = torch.zeros(n * batch + 4, dtype=torch.int32, device=dev)
            dst = torch.zeros(n * batch + 4, dtype=torch.int32, device=dev)
            src[shift:shift + n * batch] = torch.from_numpy(x.view(np.int32)).to(dev)
            assert (src.data_ptr() + 4 * shift) % 16 == 4 * shift
            N.ntt(fname, src.data_ptr() + 4 * shift, direction, cfg, out=dst.data_ptr() + 4 * shift, size=n)
            torch.cuda.synchronize()
            got = dst[shift:shift + n * batch].cpu().numpy().view(np.uint32)
            assert np.array_equal(got, exp), (direction, ordering, shift)
            assert int(dst[:shift].abs().sum()) == 0 and int(dst[shift + n * batch:].abs().sum()) == 0  # nothing outside


def test_ntt_extension_field(env, hip):
    fname, F, rf, N = env
    rng = np.random.default_rng(5)
    for logn, batch, columns in ((5, 1, False), (9, 2, False), (13, 3, True)):
        n = 1 << logn
        x = rng.integers(0, F.p, size=n * batch * 4, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.columns_batch = batch, columns
        for direction in (0, 1):
            got = N.ntt(fname, x, direction, cfg, extension=True)
            exp = rf.ntt(x, n, direction, batch=batch, columns_batch=columns, extension=True)
            assert np.array_equal(got, exp), (logn, batch, columns, direction)


def test_ntt_errors(env, hip):
    fname, F, rf, N = env
    x = np.zeros(12, dtype=np.uint32)
    with pytest.raises(hip.IcicleError):  # not a power of two
        N.ntt(fname, x, N.FORWARD, size=12)
    big = np.zeros(1 << (DOMAIN_LOG[fname] + 1), dtype=np.uint32)
    with pytest.raises(hip.IcicleError):  # larger than the domain (cpu_ntt_main.h:38-41)
        N.ntt(fname, big, N.FORWARD)


def test_ntt_roundtrip_large(env, hip):
    """size-independent property at a size the CPU oracle would take long on: inverse(forward(x)) == x,
    and linearity NTT(a+b) = NTT(a)+NTT(b)."""
    fname, F, rf, N = env
    rng = np.random.default_rng(9)
    logn, batch = DOMAIN_LOG[fname], 4
    n = 1 << logn
    a = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
    b = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
    cfg = hip.NTTConfigU32.default()
    cfg.batch_size = batch
    fa, fb = N.ntt(fname, a, N.FORWARD, cfg), N.ntt(fname, b, N.FORWARD, cfg)
    assert np.array_equal(N.ntt(fname, fa, N.INVERSE, cfg), a)
    s = ((a.astype(np.uint64) + b) % F.p).astype(np.uint32)
    fs = N.ntt(fname, s, N.FORWARD, cfg)
    assert np.array_equal(fs, ((fa.astype(np.uint64) + fb) % F.p).astype(np.uint32))
    # one row checked against the reference
    assert np.array_equal(fa[:n], rf.ntt(a[:n], n, 0))


def test_distributed_ntt_single_rank_path(env, hip):
    """world_size 1 run of the multi-GPU 4-step transform (local transposes instead of all-to-all): exercises
    the twiddle helper and the two local batched NTTs on the GPU against the plain single-call transform."""
    import torch
    from icicle_amd import dist as D

    fname, F, rf, N = env
    rng = np.random.default_rng(21)
    for logn, inverse in ((12, False), (15, True), (18, False)):
        n = 1 << logn
        x = rng.integers(0, F.p, size=n, dtype=np.uint32)
        chunk = torch.from_numpy(x.view(np.int32).copy()).cuda()
        got = D.ntt_distributed(fname, chunk, logn, inverse, 0, 1, None).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, N.ntt(fname, x, N.INVERSE if inverse else N.FORWARD))


def test_ntt_extension_field_orderings_and_cosets(env, hip):
    """the quartic-extension NTT through every ordering / coset combination (the lanes ride the same fast path)"""
    fname, F, rf, N = env
    rng = np.random.default_rng(55)
    for logn, batch in ((6, 2), (11, 1), (14, 2)):
        n = 1 << logn
        for trial in range(4):
            columns = bool(rng.integers(0, 2))
            ordering = int(rng.integers(0, 4))
            direction = int(rng.integers(0, 2))
            coset = 1 if trial == 0 else int(rng.integers(2, F.p))
            x = rng.integers(0, F.p, size=n * batch * 4, dtype=np.uint32)
            cfg = hip.NTTConfigU32.default()
            cfg.batch_size, cfg.columns_batch, cfg.ordering, cfg.coset_gen = batch, columns, ordering, coset
            got = N.ntt(fname, x, direction, cfg, extension=True)
            exp = rf.ntt(x, n, direction, batch=batch, columns_batch=columns, ordering=ordering, coset_gen=coset, extension=True)
            assert np.array_equal(got, exp), (logn, batch, columns, ordering, direction, coset)


def test_ntt_many_tiny_transforms(env, hip):
    """more rows than one grid dimension holds (70000 transforms of 4 and 256 points), every path"""
    fname, F, rf, N = env
    rng = np.random.default_rng(56)
    # (no coset at this batch size: the reference's CPU coset path takes ~20 s per 70000 rows)
    for logn, ordering, coset, batch in ((2, 2, 1, 70000), (2, 0, 5, 300), (8, 2, 7, 300), (8, 2, 1, 70000), (8, 1, 1, 70000)):
        n = 1 << logn
        x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.ordering, cfg.coset_gen = batch, ordering, coset
        got = N.ntt(fname, x, N.FORWARD, cfg)
        exp = rf.ntt(x, n, 0, batch=batch, ordering=ordering, coset_gen=coset)
        assert np.array_equal(got, exp), (logn, ordering, coset)


@pytest.mark.parametrize("G", [1, 2, 8])
def test_ntt_multi_device_through_c_abi(env, hip, G):
    """config.ext {"hip_num_devices": G}: the batch cut into G row shards behind the unchanged <field>_ntt symbol
    (logical shards share GPU 0 here; rows are independent transforms, no collective). Host and device-resident
    operands, in place, forward with coset / inverse kNR -- memcmp against the reference CPU backend."""
    import ctypes
    from icicle_amd._lib import lib
    from icicle_amd.runtime import DeviceVec

    fname, F, rf, N = env
    rng = np.random.default_rng(77 + G)
    ext = lib.create_config_extension()
    try:
        lib.config_extension_set_int(ext, b"hip_num_devices", G)
        for logn, batch, direction, ordering, coset in ((12, 5, 0, 0, 1), (9, 11, 1, 1, 1), (14, 3, 0, 0, 13), (6, 1, 0, 0, 1)):
            n = 1 << logn
            x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
            cfg = hip.NTTConfigU32.default()
            cfg.batch_size, cfg.ordering, cfg.coset_gen, cfg.ext = batch, ordering, coset, ext
            exp = rf.ntt(x, n, direction, batch=batch, ordering=ordering, coset_gen=coset)
            assert np.array_equal(N.ntt(fname, x, direction, cfg), exp), (G, logn, batch, "host")
            d = DeviceVec.from_host(x)
            N.ntt(fname, d, direction, cfg, out=d, size=n)  # device resident, in place
            assert np.array_equal(d.to_host(), exp), (G, logn, batch, "device in place")
    finally:
        lib.destroy_config_extension(ext)


def test_stream_ordered_alloc_and_workspace(env, hip):
    """operands from icicle_malloc_async / icicle_free_async (icicle/tests/test_device_api.cpp:98-118 pattern) through an
    NTT and an MSM on a created stream; then the cached workspace is reported and given back."""
    import ctypes
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check
    from icicle_amd.runtime import Stream
    from tests.util import cached_points, points_to_array, rand_scalars, to_words

    fname, F, rf, N = env
    rng = np.random.default_rng(88)
    st = Stream()
    n = 1 << 13
    x = rng.integers(0, F.p, size=n, dtype=np.uint32)
    p_in, p_out = ctypes.c_void_p(), ctypes.c_void_p()
    check(lib.icicle_malloc_async(ctypes.byref(p_in), x.nbytes, st.handle))
    check(lib.icicle_malloc_async(ctypes.byref(p_out), x.nbytes, st.handle))
    assert lib.icicle_is_active_device_memory(p_in) == 0 and lib.icicle_is_host_memory(p_in) != 0
    check(lib.icicle_copy_to_device_async(p_in, x.ctypes.data, x.nbytes, st.handle))
    cfg = hip.NTTConfigU32.default()
    cfg.stream, cfg.is_async = st.handle, True
    N.ntt(fname, p_in.value, N.FORWARD, cfg, out=p_out.value, size=n)
    y = np.empty_like(x)
    check(lib.icicle_copy_to_host_async(y.ctypes.data, p_out, x.nbytes, st.handle))
    check(lib.icicle_free_async(p_in, st.handle))
    check(lib.icicle_free_async(p_out, st.handle))
    st.synchronize()
    assert np.array_equal(y, rf.ntt(x, n, 0))
    assert lib.icicle_is_host_memory(p_in) == 0  # no longer tracked as device memory
    # MSM on malloc_async'd operands
    C = pyref.BN254
    refc = ref.RefCurve("bn254")
    m = 3000
    bases = points_to_array(C, cached_points(C, m))
    sc = to_words(rand_scalars(rng, m, C.r), 8)
    ps, pb, pr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    for ptr, nb in ((ps, sc.nbytes), (pb, bases.nbytes), (pr, 96)):
        check(lib.icicle_malloc_async(ctypes.byref(ptr), nb, st.handle))
    check(lib.icicle_copy_to_device_async(ps, sc.ctypes.data, sc.nbytes, st.handle))
    check(lib.icicle_copy_to_device_async(pb, bases.ctypes.data, bases.nbytes, st.handle))
    mc = hip.MSMConfig.default()
    mc.stream, mc.is_async = st.handle, True
    M.msm("bn254", ps.value, pb.value, mc, results=pr.value, msm_size=m)
    out = np.zeros((1, 24), dtype=np.uint32)
    check(lib.icicle_copy_to_host_async(out.ctypes.data, pr, 96, st.handle))
    for ptr in (ps, pb, pr):
        check(lib.icicle_free_async(ptr, st.handle))
    st.synchronize()
    assert np.array_equal(refc.to_affine(out), refc.to_affine(refc.msm(sc, bases)))
    # 200 live allocations, then all released (test_device_api.cpp:191-254 counts live allocations the same way)
    ptrs = []
    for i in range(200):
        q = ctypes.c_void_p()
        check(lib.icicle_malloc(ctypes.byref(q), 1024 + i))
        ptrs.append(q)
    assert len({q.value for q in ptrs}) == 200
    for q in ptrs:
        check(lib.icicle_free(q))
    cached = ctypes.c_size_t()
    check(lib.icicle_hip_workspace_bytes(ctypes.byref(cached)))
    assert cached.value > 0
    check(lib.icicle_hip_release_workspace())
    check(lib.icicle_hip_workspace_bytes(ctypes.byref(cached)))
    assert cached.value == 0
    st.destroy()


@pytest.mark.parametrize("logn", list(range(1, 21)))
def test_ntt_bit_reversed_input_consumed_natively(env, hip, logn):
    """Round 5: kRN without the reordering pre-pass (ntt_fast.hpp RN: a run pass over the contiguous reversed runs, then
    in-place column passes; the reference reads through a permutation at no extra data pass either, ntt_cpu.h:252,286-296).
    Every size 2^1 .. 2^20 (1, 2 and 3 passes, every sub-transform length), both directions, inverse with a coset, row
    batches, columns_batch (lane-native tiles, full and ragged slices), the extension field, in place -- memcmp with the
    reference CPU backend. Forward cosets and kRR (which keep the pre-pass) ride along as controls."""
    import ctypes

    fname, F, rf, N = env
    rng = np.random.default_rng(7000 + logn)
    n = 1 << logn
    cases = [(1, False, N.INVERSE, 1), (3, False, N.FORWARD, 1), (5, False, N.INVERSE, int(rng.integers(2, F.p))),
             (32, True, N.INVERSE, 1), (64, True, N.FORWARD, 1), (7, True, N.FORWARD, 1), (37, True, N.INVERSE, int(rng.integers(2, F.p))),
             (2, False, N.FORWARD, int(rng.integers(2, F.p)))]
    if logn > 16:
        cases = cases[:2] + [cases[3], cases[4], cases[6]]
    for batch, columns, direction, coset in cases:
        x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.columns_batch, cfg.ordering, cfg.coset_gen = batch, columns, N.kRN, coset
        exp = rf.ntt(x, n, direction, batch=batch, columns_batch=columns, ordering=N.kRN, coset_gen=coset)
        got = N.ntt(fname, x, direction, cfg)
        assert np.array_equal(got, exp), (logn, batch, columns, direction, coset)
        y = x.copy()  # in place (host buffers: staged once, transformed in place on the device)
        N.ntt(fname, y, direction, cfg, out=y)
        assert np.array_equal(y, exp), ("in place", logn, batch, columns, direction, coset)
    # kRR control + the extension field (4 interleaved base-field transforms per element)
    x = rng.integers(0, F.p, size=n * 2, dtype=np.uint32)
    cfg = hip.NTTConfigU32.default()
    cfg.batch_size, cfg.ordering = 2, N.kRR
    assert np.array_equal(N.ntt(fname, x, N.FORWARD, cfg), rf.ntt(x, n, 0, batch=2, ordering=N.kRR))
    if logn <= 16:
        xe = rng.integers(0, F.p, size=n * 4 * 3, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.ordering = 3, N.kRN
        for direction in (N.FORWARD, N.INVERSE):
            assert np.array_equal(N.ntt(fname, xe, direction, cfg, extension=True), rf.ntt(xe, n, direction, batch=3, ordering=N.kRN, extension=True)), ("ext", logn, direction)
