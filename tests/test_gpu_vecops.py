"""GPU parity: Montgomery-form conversion of scalars and points vs the reference
CPU backend, memcmp-exact, both directions, host and device buffers; and the wrapper flow it exists for:
scalars converted on the device, then msm(are_scalars_montgomery_form=True)
(wrappers/rust/icicle-core/src/msm/tests.rs:54-59)."""
import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import cached_points, points_to_array, rand_scalars, to_words

pytestmark = pytest.mark.gpu
BIG = ("bn254", "bls12_381", "bls12_377", "grumpkin", "stark252")  # 8-word scalars


def scalar_modulus(field):
    """a curve name stands for its scalar field"""
    return pyref.CURVES[field].r if field in pyref.CURVES else pyref.NTT_FIELDS[field].p


@pytest.mark.parametrize("field", ["bn254", "bls12_381", "bls12_377", "grumpkin", "stark252", "babybear", "koalabear"])
def test_scalar_convert_montgomery(hip, field):
    from icicle_amd import vecops as V
    from icicle_amd.runtime import DeviceVec

    rng = np.random.default_rng(3)
    n = 5000
    if field in BIG:
        p = scalar_modulus(field)
        vals = rand_scalars(rng, n, p)
        vals[:4] = [0, 1, p - 1, p // 2]
        x = to_words(vals, 8)
    else:
        p = pyref.NTT_FIELDS[field].p
        x = rng.integers(0, p, size=n, dtype=np.uint32)
        x[:3] = [0, 1, p - 1]
    sym = f"{field}_scalar_convert_montgomery"
    to_ref = ref.ref_convert_montgomery(field, sym, x, n, True)
    got = V.scalar_convert_montgomery(field, x, True, size=n)
    assert np.array_equal(got, to_ref)
    assert np.array_equal(V.scalar_convert_montgomery(field, got, False, size=n), x)
    assert np.array_equal(ref.ref_convert_montgomery(field, sym, got, n, False), x)
    # device-resident, in place
    d = DeviceVec.from_host(x)
    V.scalar_convert_montgomery(field, d, True, out=d, size=n)
    assert np.array_equal(d.to_host(shape=x.shape), to_ref)
    if field in ("babybear", "koalabear"):
        xe = rng.integers(0, p, size=4 * 300, dtype=np.uint32)
        ge = V.scalar_convert_montgomery(field, xe, True, size=300, extension=True)
        assert np.array_equal(ge, ref.ref_convert_montgomery(field, f"{field}_extension_scalar_convert_montgomery", xe, 300, True))


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377", "grumpkin"])
def test_point_convert_montgomery_and_wrapper_flow(hip, cname):
    from icicle_amd import msm as M
    from icicle_amd import vecops as V

    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(5)
    n = 1200
    pts = list(cached_points(C, n))
    pts[3] = pyref.INF
    aff = points_to_array(C, pts)
    am = V.affine_convert_montgomery(cname, aff, True)
    assert np.array_equal(am, ref.ref_convert_montgomery(cname, f"{cname}_affine_convert_montgomery", aff, n, True))
    assert np.array_equal(V.affine_convert_montgomery(cname, am, False), aff)
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    plain = M.msm(cname, sc, aff)
    pm = V.projective_convert_montgomery(cname, plain, True)
    assert np.array_equal(pm, ref.ref_convert_montgomery(cname, f"{cname}_projective_convert_montgomery", plain, 1, True))
    # the Rust test flow: Montgomery scalars (converted on device) + Montgomery points
    scm = V.scalar_convert_montgomery(cname, sc, True, size=n)
    cfg = hip.MSMConfig.default()
    cfg.are_scalars_montgomery_form = True
    cfg.are_points_montgomery_form = True
    got = M.msm(cname, scm, am, cfg)
    assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, aff)))


def _ref_vec2(fname, op, a, b, size, batch=1, columns=False):
    """<field>_<op> of the reference on its CPU device (src/vec_ops.cpp)"""
    import ctypes

    from oracle.ref import REF_DIR, VecOpsConfig
    import os

    lib = ctypes.CDLL(os.path.join(REF_DIR, f"libicicle_field_{fname}.so"))
    cfg = VecOpsConfig(None, False, False, False, False, batch, columns, None)
    out = np.zeros_like(b)
    fn = getattr(lib, f"{fname}_{op}")
    if op == "bit_reverse":
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        assert fn(b.ctypes.data, size, ctypes.byref(cfg), out.ctypes.data) == 0
    else:
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        assert fn(a.ctypes.data, b.ctypes.data, size, ctypes.byref(cfg), out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("fname", ["babybear", "koalabear", "bn254", "bls12_381", "bls12_377", "grumpkin", "stark252"])
def test_vector_arithmetic_vs_reference(hip, fname):
    """vector_add / sub / mul, scalar_mul_vec, bit_reverse vs the reference CPU backend (memcmp), with batches in
    both layouts, edge values (0, 1, p-1) and device-resident operands"""
    from icicle_amd import vecops as V
    from icicle_amd.runtime import DeviceVec

    P = scalar_modulus(fname)
    W = 8 if fname in BIG else 1
    rng = np.random.default_rng(71)

    def rand(count):
        vals = rand_scalars(rng, count, P)
        vals[:3] = [0, 1, P - 1][: min(3, count)]
        return np.ascontiguousarray(to_words(vals, W).reshape(-1))

    for size, batch, columns in ((1, 1, False), (1000, 1, False), (256, 3, False), (256, 3, True), (1 << 14, 2, True)):
        a, b = rand(size * batch), rand(size * batch)
        cfg = hip.VecOpsConfig.default()
        cfg.batch_size, cfg.columns_batch = batch, columns
        for op, fn in (("vector_add", V.vector_add), ("vector_sub", V.vector_sub), ("vector_mul", V.vector_mul)):
            assert np.array_equal(fn(fname, a, b, cfg), _ref_vec2(fname, op, a, b, size, batch, columns)), (fname, op, size, batch, columns)
        s = rand(batch)
        assert np.array_equal(V.scalar_mul_vec(fname, s, b, cfg), _ref_vec2(fname, "scalar_mul_vec", s, b, size, batch, columns))
        if size & (size - 1) == 0:
            assert np.array_equal(V.bit_reverse(fname, b, cfg), _ref_vec2(fname, "bit_reverse", None, b, size, batch, columns))
    # device-resident, async on the default stream, in place for bit_reverse
    size = 1 << 12
    a, b = rand(size), rand(size)
    da, db = DeviceVec.from_host(a), DeviceVec.from_host(b)
    cfg = hip.VecOpsConfig.default()
    V.vector_mul(fname, da, db, cfg, out=db, size=size)
    assert np.array_equal(db.to_host(), _ref_vec2(fname, "vector_mul", a, b, size))
    V.bit_reverse(fname, da, hip.VecOpsConfig.default(), out=da, size=size)
    assert np.array_equal(da.to_host(), _ref_vec2(fname, "bit_reverse", None, a, size))


def test_polynomial_product_pipeline_on_device(hip):
    """NTT(kNR) -> point-wise product -> inverse NTT(kRN), all device resident: (1 + 2x)(3 + x) and a random pair"""
    from icicle_amd import ntt as N
    from icicle_amd import vecops as V
    from icicle_amd.runtime import DeviceVec

    F = pyref.BABYBEAR
    logn = 12
    n = 1 << logn
    N.init_domain("babybear", N.get_root_of_unity("babybear", n))
    try:
        rng = np.random.default_rng(3)
        fa = np.zeros(n, dtype=np.uint32)
        fb = np.zeros(n, dtype=np.uint32)
        fa[: n // 2] = rng.integers(0, F.p, size=n // 2, dtype=np.uint32)
        fb[: n // 2] = rng.integers(0, F.p, size=n // 2, dtype=np.uint32)
        da, db = DeviceVec.from_host(fa), DeviceVec.from_host(fb)
        cf = hip.NTTConfigU32.default()
        cf.ordering = N.kNR
        N.ntt("babybear", da, N.FORWARD, cf, out=da, size=n)
        N.ntt("babybear", db, N.FORWARD, cf, out=db, size=n)
        V.vector_mul("babybear", da, db, hip.VecOpsConfig.default(), out=da, size=n)
        ci = hip.NTTConfigU32.default()
        ci.ordering = N.kRN
        N.ntt("babybear", da, N.INVERSE, ci, out=da, size=n)
        got = da.to_host()
        exp = np.zeros(n, dtype=object)
        A = [int(v) for v in fa[: n // 2]]
        B = [int(v) for v in fb[: n // 2]]
        conv = np.convolve(np.array(A, dtype=object), np.array(B, dtype=object))
        exp[: conv.size] = conv
        assert [int(v) for v in got] == [int(v) % F.p for v in exp]
    finally:
        N.release_domain("babybear")


@pytest.mark.parametrize("field,words,ext_words", [("babybear", 1, 4), ("koalabear", 1, 4), ("goldilocks", 2, 4), ("bn254", 8, 0), ("stark252", 8, 0)])
def test_matrix_transpose(hip, field, words, ext_words):
    """<field>_matrix_transpose / _extension_matrix_transpose through the C ABI (icicle/src/matrix_ops.cpp:75-102): batches of
    ragged and tile-aligned matrices, host and device operands, in place (cpu_matrix_ops.cpp:348-359), and the argument errors
    of cpu_matrix_ops.cpp:337-345. Pure data movement: the expected bytes are numpy's transpose of the same words (the reference
    CPU backend's own out-of-place loop, cpu_matrix_ops.cpp:184-196, is that definition; it is run on the same inputs through the
    reference runtime in tests/plugin_driver.py)."""
    from icicle_amd import vecops as V
    from icicle_amd.runtime import DeviceVec

    rng = np.random.default_rng(17)
    for w, ext in ((words, False),) + (((ext_words, True),) if ext_words else ()):
        for rows, cols, batch in ((1, 1, 1), (37, 200, 3), (128, 256, 3), (1000, 33, 2), (32, 32, 70000 if w == 1 else 5), (4096, 100, 1)):
            x = rng.integers(0, 1 << 32, size=(batch, rows, cols, w), dtype=np.uint64).astype(np.uint32)
            exp = np.ascontiguousarray(x.transpose(0, 2, 1, 3))
            cfg = hip.VecOpsConfig.default()
            cfg.batch_size = batch
            got = V.matrix_transpose(field, x.reshape(-1), rows, cols, cfg, extension=ext)
            assert np.array_equal(got, exp.reshape(-1)), (field, ext, rows, cols, batch)
        # device operands: out of place, then in place
        rows, cols, batch = 300, 77, 4
        x = rng.integers(0, 1 << 32, size=(batch, rows, cols, w), dtype=np.uint64).astype(np.uint32)
        exp = np.ascontiguousarray(x.transpose(0, 2, 1, 3)).reshape(-1)
        cfg = hip.VecOpsConfig.default()
        cfg.batch_size = batch
        d_in, d_out = DeviceVec.from_host(x.reshape(-1)), DeviceVec.from_host(np.zeros(x.size, dtype=np.uint32))
        V.matrix_transpose(field, d_in, rows, cols, cfg, out=d_out, extension=ext)
        assert np.array_equal(d_out.to_host(), exp)
        V.matrix_transpose(field, d_in, rows, cols, cfg, out=d_in, extension=ext)
        assert np.array_equal(d_in.to_host(), exp)
    x = np.zeros(64 * words, dtype=np.uint32)
    cfg = hip.VecOpsConfig.default()
    cfg.columns_batch = True
    with pytest.raises(hip.IcicleError):  # cpu_matrix_ops.cpp:342-345
        V.matrix_transpose(field, x, 8, 8, cfg)
    with pytest.raises(hip.IcicleError):  # :337-340
        V.matrix_transpose(field, x, 0, 8)
