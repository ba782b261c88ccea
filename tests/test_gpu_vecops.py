"""GPU parity: Montgomery-form conversion of scalars and points vs the reference
CPU backend, memcmp-exact, both directions, host and device buffers; and the wrapper flow it exists for:
scalars converted on the device, then msm(are_scalars_montgomery_form=True)
(wrappers/rust/icicle-core/src/msm/tests.rs:54-59)."""
import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import cached_points, points_to_array, rand_scalars, to_words

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field", ["bn254", "bls12_381", "babybear", "koalabear"])
def test_scalar_convert_montgomery(hip, field):
    from icicle_amd import vecops as V
    from icicle_amd.runtime import DeviceVec

    rng = np.random.default_rng(3)
    n = 5000
    if field in ("bn254", "bls12_381"):
        p = pyref.CURVES[field].r
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


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
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
