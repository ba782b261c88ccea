"""GPU parity: NTT over the 256-bit fields (the curves' scalar fields <curve>_ntt, and stark252_ntt) through the C ABI
vs the reference CPU backend, memcmp-exact. Same random matrix as test_gpu_ntt.py
(icicle/tests/test_mod_arithmetic_api.h:614-695: logn, batch, columns_batch, direction, ordering, coset)."""
import ctypes

import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu
FIELDS = ["bn254", "bls12_381", "bls12_377", "stark252"]
DOMAIN_LOG = 18


def rand_elems(rng, p: int, count: int) -> np.ndarray:
    """count canonical field elements as [count*8] u32 words"""
    raw = rng.integers(0, 1 << 32, size=(count, 8), dtype=np.uint64)
    vals = [sum(int(raw[i, j]) << (32 * j) for j in range(8)) % p for i in range(count)]
    out = np.array([[(v >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for v in vals], dtype=np.uint32)
    return out.reshape(-1)


def to_ints(words: np.ndarray):
    w = words.reshape(-1, 8)
    return [sum(int(w[i, j]) << (32 * j) for j in range(8)) for i in range(w.shape[0])]


@pytest.fixture(scope="module", params=FIELDS)
def env(request, hip):
    from icicle_amd import ntt as N

    fname = request.param
    F = pyref.NTT_FIELDS[fname]
    rf = ref.RefScalarNttField(fname)
    root = N.get_root_of_unity(fname, 1 << DOMAIN_LOG)
    assert root == rf.get_root_of_unity(1 << DOMAIN_LOG) == pyref.omega(F, DOMAIN_LOG)
    N.init_domain(fname, root)
    N.init_domain(fname, root)  # second init is a silent success (cpu_ntt_domain.h:69)
    rf.init_domain(root)
    yield fname, F, rf, N
    N.release_domain(fname)
    rf.release_domain()


def test_rou(env):
    fname, F, rf, N = env
    for logn in (0, 1, 5, DOMAIN_LOG):
        assert N.get_root_of_unity_from_domain(fname, logn) == rf.get_root_of_unity_from_domain(logn) == pyref.omega(F, logn)
    assert N.get_root_of_unity(fname, 1) == 1
    if F.two_adicity < 63:  # (stark252: 2^192 | p - 1, max_size is a uint64)
        assert N.get_root_of_unity(fname, 1 << F.two_adicity) == F.rou
        from icicle_amd._lib import IcicleError

        with pytest.raises(IcicleError):
            N.get_root_of_unity(fname, 1 << (F.two_adicity + 1))
    else:
        assert N.get_root_of_unity(fname, 1 << 63) == rf.get_root_of_unity(1 << 63) == pyref.omega(F, 63)


def test_vs_python_definition(env, hip):
    fname, F, rf, N = env
    rng = np.random.default_rng(7)
    for logn in (0, 1, 2, 5, 7):
        n = 1 << logn
        x = rand_elems(rng, F.p, n)
        y = N.ntt(fname, x, N.FORWARD)
        assert to_ints(y) == pyref.ntt_naive(F, to_ints(x), pyref.omega(F, logn))
        back = N.ntt(fname, y, N.INVERSE)
        assert np.array_equal(back, x)


@pytest.mark.parametrize("logn", [0, 1, 3, 6, 8, 9, 11, 13, 16, 17])
def test_matrix_vs_reference(env, hip, logn):
    fname, F, rf, N = env
    rng = np.random.default_rng(2000 + logn)
    n = 1 << logn
    for trial in range(4 if logn < 16 else 2):
        batch = int(rng.choice([1, 2, 3])) if logn < 16 else 1
        columns = bool(rng.integers(0, 2))
        ordering = int(rng.integers(0, 6))
        direction = int(rng.integers(0, 2))
        coset = 1 if rng.integers(0, 2) else to_ints(rand_elems(rng, F.p, 1))[0] or 5
        x = rand_elems(rng, F.p, n * batch)
        cfg = hip.NTTConfigU256.default()
        cfg.batch_size, cfg.columns_batch, cfg.ordering = batch, columns, ordering
        cfg.set_coset_gen(coset)
        got = N.ntt(fname, x, direction, cfg)
        exp = rf.ntt(x, n, direction, batch=batch, columns_batch=columns, ordering=ordering, coset_gen=coset)
        assert np.array_equal(got, exp), (fname, logn, batch, columns, ordering, direction, hex(coset))


def test_edge_values_and_montgomery_linearity(env, hip):
    """p-1, 0, 1 inputs; and NTT(x*R) == NTT(x)*R (the transform is linear, so Montgomery-form inputs work)"""
    fname, F, rf, N = env
    n = 64
    vals = [F.p - 1, 0, 1, F.p - 2] * (n // 4)
    x = np.array([[(v >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for v in vals], dtype=np.uint32).reshape(-1)
    got = N.ntt(fname, x, N.FORWARD)
    assert np.array_equal(got, rf.ntt(x, n, 0))
    R = 1 << 256
    xm = [(v * R) % F.p for v in vals]
    xmw = np.array([[(v >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for v in xm], dtype=np.uint32).reshape(-1)
    gm = to_ints(N.ntt(fname, xmw, N.FORWARD))
    assert gm == [(v * R) % F.p for v in to_ints(got)]


def test_device_inplace_async(env, hip):
    fname, F, rf, N = env
    from icicle_amd.runtime import DeviceVec, Stream

    rng = np.random.default_rng(11)
    for logn in (5, 12, 17):
        n = 1 << logn
        x = rand_elems(rng, F.p, n)
        d = DeviceVec.from_host(x)
        st = Stream()
        cfg = hip.NTTConfigU256.default()
        cfg.stream = st.handle
        cfg.is_async = True
        N.ntt(fname, d, N.FORWARD, cfg, out=d, size=n)
        st.synchronize()
        assert np.array_equal(d.to_host(), rf.ntt(x, n, 0))
        N.ntt(fname, d, N.INVERSE, cfg, out=d, size=n)
        st.synchronize()
        assert np.array_equal(d.to_host(), x)
        st.destroy()


def test_errors(env, hip):
    fname, F, rf, N = env
    from icicle_amd._lib import IcicleError

    x = np.zeros(8 * 12, dtype=np.uint32)
    with pytest.raises(IcicleError):
        N.ntt(fname, x, N.FORWARD, size=12)  # not a power of two (cpu_ntt_main.h:38-41)
    big = np.zeros(8, dtype=np.uint32)
    with pytest.raises(IcicleError):
        N.ntt(fname, big, N.FORWARD, size=1 << (DOMAIN_LOG + 1))  # larger than the domain
    cfg = hip.NTTConfigU256.default()
    cfg.set_coset_gen(0)
    with pytest.raises(IcicleError):
        N.ntt(fname, np.zeros(8 * 4, dtype=np.uint32), N.FORWARD, cfg)


def test_roundtrip_large(env, hip):
    """2^18 x 4: forward then inverse is the identity; linearity checksum NTT(x)[0] = sum(x)"""
    fname, F, rf, N = env
    rng = np.random.default_rng(5)
    n, batch = 1 << DOMAIN_LOG, 4
    x = rand_elems(rng, F.p, 1024)
    x = np.tile(x.reshape(1024, 8), (n * batch // 1024, 1)).reshape(-1).copy()
    cfg = hip.NTTConfigU256.default()
    cfg.batch_size = batch
    y = N.ntt(fname, x, N.FORWARD, cfg)
    s = sum(to_ints(x[: n * 8])) % F.p
    assert to_ints(y[:8])[0] == s
    back = N.ntt(fname, y, N.INVERSE, cfg)
    assert np.array_equal(back, x)


def test_many_tiny_transforms(env, hip):
    """more rows than one grid dimension holds"""
    fname, F, rf, N = env
    rng = np.random.default_rng(57)
    n, batch = 4, 66000
    x = np.tile(rand_elems(rng, F.p, 1000 * n).reshape(1000 * n, 8), (batch // 1000, 1)).reshape(-1).copy()
    cfg = hip.NTTConfigU256.default()
    cfg.batch_size, cfg.ordering = batch, 2
    got = N.ntt(fname, x, N.FORWARD, cfg)
    exp = rf.ntt(x, n, 0, batch=batch, ordering=2)
    assert np.array_equal(got, exp)
