"""GPU parity: goldilocks_ntt / goldilocks_extension_ntt (8-byte elements; the quadratic extension transforms its two
components with the base field's twiddles), Montgomery conversion and the vector ops through the C ABI vs the reference CPU
backend built for FIELD_ID 1005, memcmp-exact -- the same random matrix as tests/test_gpu_ntt.py
(icicle/tests/test_mod_arithmetic_api.h:614-695: logn, batch, columns_batch, direction, ordering, coset)."""
import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu
F = pyref.GOLDILOCKS
DOMAIN_LOG = 20
FNAME = "goldilocks"


def rand_elems(rng, count):
    """count canonical elements as [count*2] u32 words"""
    v = rng.integers(0, 1 << 63, size=count, dtype=np.uint64) * 2 + rng.integers(0, 2, size=count, dtype=np.uint64)
    v = np.where(v >= np.uint64(F.p), v - np.uint64(F.p), v)
    return np.ascontiguousarray(v.astype("<u8").view(np.uint32))


def to_ints(words):
    return [int(x) for x in np.ascontiguousarray(words).view("<u8")]


@pytest.fixture(scope="module")
def env(hip):
    from icicle_amd import ntt as N

    rf = ref.RefGoldField()
    root = N.get_root_of_unity(FNAME, 1 << DOMAIN_LOG)
    assert root == rf.get_root_of_unity(1 << DOMAIN_LOG) == pyref.omega(F, DOMAIN_LOG)
    N.init_domain(FNAME, root)
    N.init_domain(FNAME, root)  # second init is a silent success (cpu_ntt_domain.h:69)
    rf.init_domain(root)
    yield rf, N
    N.release_domain(FNAME)
    rf.release_domain()


def test_roots(env):
    rf, N = env
    for logn in (0, 1, 5, DOMAIN_LOG):
        assert N.get_root_of_unity_from_domain(FNAME, logn) == rf.get_root_of_unity_from_domain(logn) == pyref.omega(F, logn)
    assert N.get_root_of_unity(FNAME, 1) == 1
    assert N.get_root_of_unity(FNAME, 1 << F.two_adicity) == F.rou
    from icicle_amd._lib import IcicleError

    with pytest.raises(IcicleError):
        N.get_root_of_unity(FNAME, 1 << (F.two_adicity + 1))


def test_vs_python_definition(env, hip):
    rf, N = env
    rng = np.random.default_rng(1)
    for logn in (0, 1, 2, 5, 8):
        n = 1 << logn
        x = rand_elems(rng, n)
        if logn == 5:  # edge values
            x.view("<u8")[:4] = [0, 1, F.p - 1, (1 << 32) - 1]
        y = N.ntt(FNAME, x, N.FORWARD)
        assert to_ints(y) == pyref.ntt_naive(F, to_ints(x), pyref.omega(F, logn))
        assert np.array_equal(N.ntt(FNAME, y, N.INVERSE), x)


@pytest.mark.parametrize("logn", [0, 1, 3, 6, 8, 9, 12, 13, 16, 17])
def test_matrix_vs_reference(env, hip, logn):
    rf, N = env
    rng = np.random.default_rng(2000 + logn)
    n = 1 << logn
    for trial in range(6):
        batch = int(rng.choice([1, 2, 4, 7]))
        columns = bool(rng.integers(0, 2))
        ordering = int(rng.integers(0, 6))
        direction = int(rng.integers(0, 2))
        coset = 1 if rng.integers(0, 2) else int(rng.integers(2, 1 << 62))
        ext = bool(rng.integers(0, 2))
        x = rand_elems(rng, n * batch * (2 if ext else 1))
        cfg = hip.NTTConfigU64.default()
        cfg.batch_size, cfg.columns_batch, cfg.ordering = batch, columns, ordering
        cfg.set_coset_gen(coset)
        got = N.ntt(FNAME, x, direction, cfg, extension=ext)
        exp = rf.ntt(x, n, direction, batch=batch, columns_batch=columns, ordering=ordering, coset_gen=coset, extension=ext)
        assert np.array_equal(got, exp), (logn, batch, columns, ordering, direction, coset, ext)


def test_extension_is_the_base_transform_per_component(env, hip):
    rf, N = env
    rng = np.random.default_rng(5)
    n = 1 << 10
    x = rand_elems(rng, 2 * n)
    y = N.ntt(FNAME, x, N.FORWARD, extension=True)
    c = x.reshape(n, 2, 2)
    for k in range(2):
        comp = np.ascontiguousarray(c[:, k, :]).reshape(-1)
        assert np.array_equal(np.ascontiguousarray(y.reshape(n, 2, 2)[:, k, :]).reshape(-1), N.ntt(FNAME, comp, N.FORWARD))


def test_device_inplace_async_and_large(env, hip):
    rf, N = env
    from icicle_amd.runtime import DeviceVec, Stream

    rng = np.random.default_rng(77)
    for logn, batch in ((11, 3), (18, 2), (20, 1)):
        n = 1 << logn
        x = rand_elems(rng, n * batch)
        d = DeviceVec.from_host(x)
        st = Stream()
        cfg = hip.NTTConfigU64.default()
        cfg.batch_size, cfg.stream, cfg.is_async = batch, st.handle, True
        N.ntt(FNAME, d, N.FORWARD, cfg, out=d, size=n)  # in place on device
        st.synchronize()
        assert np.array_equal(d.to_host(), rf.ntt(x, n, 0, batch=batch))
        N.ntt(FNAME, d, N.INVERSE, cfg, out=d, size=n)
        st.synchronize()
        assert np.array_equal(d.to_host(), x)
        st.destroy()


def test_errors(env, hip):
    rf, N = env
    from icicle_amd._lib import IcicleError

    with pytest.raises(IcicleError):
        N.ntt(FNAME, np.zeros(2 * 12, dtype=np.uint32), N.FORWARD, size=12)  # not a power of two
    with pytest.raises(IcicleError):
        N.ntt(FNAME, np.zeros(2, dtype=np.uint32), N.FORWARD, size=1 << (DOMAIN_LOG + 1))  # larger than the domain
    cfg = hip.NTTConfigU64.default()
    cfg.set_coset_gen(0)
    with pytest.raises(IcicleError):
        N.ntt(FNAME, np.zeros(2 * 4, dtype=np.uint32), N.FORWARD, cfg)


def test_montgomery_conversion_and_vector_ops(hip):
    from icicle_amd import vecops as V
    from tests.test_gpu_vecops import _ref_vec2

    rng = np.random.default_rng(9)
    n = 5000
    x = rand_elems(rng, n)
    x.view("<u8")[:4] = [0, 1, F.p - 1, (1 << 32) - 1]
    sym = "goldilocks_scalar_convert_montgomery"
    to_ref = ref.ref_convert_montgomery(FNAME, sym, x, n, True)
    assert to_ints(to_ref)[:3] == [0, (1 << 64) % F.p, (F.p - 1) * (1 << 64) % F.p]  # x * 2^64 (goldilocks.h:179)
    got = V.scalar_convert_montgomery(FNAME, x, True, size=n)
    assert np.array_equal(got, to_ref)
    assert np.array_equal(V.scalar_convert_montgomery(FNAME, got, False, size=n), x)
    xe = rand_elems(rng, 2 * 300)
    ge = V.scalar_convert_montgomery(FNAME, xe, True, size=300, extension=True)
    assert np.array_equal(ge, ref.ref_convert_montgomery(FNAME, "goldilocks_extension_scalar_convert_montgomery", xe, 300, True))
    for size, batch, columns in ((1, 1, False), (1000, 1, False), (256, 3, True), (1 << 14, 2, False)):
        a, b = rand_elems(rng, size * batch), rand_elems(rng, size * batch)
        cfg = hip.VecOpsConfig.default()
        cfg.batch_size, cfg.columns_batch = batch, columns
        for op, fn in (("vector_add", V.vector_add), ("vector_sub", V.vector_sub), ("vector_mul", V.vector_mul)):
            assert np.array_equal(fn(FNAME, a, b, cfg), _ref_vec2(FNAME, op, a, b, size, batch, columns)), (op, size, batch, columns)
        s = rand_elems(rng, batch)
        assert np.array_equal(V.scalar_mul_vec(FNAME, s, b, cfg), _ref_vec2(FNAME, "scalar_mul_vec", s, b, size, batch, columns))
        if size & (size - 1) == 0:
            assert np.array_equal(V.bit_reverse(FNAME, b, cfg), _ref_vec2(FNAME, "bit_reverse", None, b, size, batch, columns))


def test_golden(hip):
    import os

    from icicle_amd import ntt as N

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ntt_goldilocks.npz"))
    N.init_domain(FNAME, int(g["domain_root"].view("<u8")[0]))
    try:
        cg = int(g["coset_gen"].view("<u8")[0])

        def run(direction, x=None, ext=False, **kw):
            cfg = hip.NTTConfigU64.default()
            cfg.batch_size = kw.get("batch", 2)
            cfg.ordering = kw.get("ordering", 0)
            cfg.columns_batch = kw.get("columns", False)
            cfg.set_coset_gen(kw.get("coset", 1))
            return N.ntt(FNAME, g["x"] if x is None else x, direction, cfg, size=kw.get("size", 1024), extension=ext)

        assert np.array_equal(run(0), g["fwd_NN"])
        assert np.array_equal(run(1), g["inv_NN"])
        assert np.array_equal(run(0, ordering=1, coset=cg), g["fwd_NR_coset"])
        assert np.array_equal(run(1, ordering=2, coset=cg), g["inv_RN_coset"])
        assert np.array_equal(run(0, ordering=3), g["fwd_RR"])
        assert np.array_equal(run(0, columns=True), g["fwd_columns"])
        assert np.array_equal(run(0, x=g["x_ext"], ext=True, batch=1, size=64), g["fwd_ext"])
    finally:
        N.release_domain(FNAME)
