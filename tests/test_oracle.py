"""CPU: pin the oracle. The C restatement (oracle/oracle.c) and -- when present -- the real reference
build (oracle/_ref) must both reproduce the committed golden fixtures (minted from oracle/_ref by
tests/golden/make_golden.py), and the pure-Python definitions must agree on small cases."""
import os

import numpy as np
import pytest

from oracle import pyref, ref
from tests import oracle_c as oc
from tests.util import from_words, points_to_array, rand_scalars, to_words

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CURVES = ["bn254", "bls12_381", "bls12_377", "grumpkin"]
FIELDS = ["babybear", "koalabear"]
ORD = {"NN": 0, "NR": 1, "RN": 2, "RR": 3}


@pytest.mark.parametrize("cname", CURVES)
def test_c_oracle_msm_matches_golden(cname):
    g = np.load(os.path.join(GOLD, f"msm_{cname}.npz"))
    n = g["bases"].shape[0]
    for c in (7, 12):
        got = oc.to_affine(cname, oc.msm(cname, np.ascontiguousarray(g["scalars"][:n]), g["bases"], c=c))
        assert np.array_equal(got, g["res_single"][0]), c
    got = oc.to_affine(cname, oc.msm(cname, np.ascontiguousarray(g["scalars"][n:]), g["bases"], c=9))
    assert np.array_equal(got, g["res_batch2_shared"][1])
    got = oc.to_affine(cname, oc.msm(cname, g["scalars_20bit"], g["bases"], c=6, bitsize=20))
    assert np.array_equal(got, g["res_bitsize20"][0])


@pytest.mark.parametrize("cname", CURVES)
def test_reference_build_matches_golden(cname):
    if not ref.available(cname):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    g = np.load(os.path.join(GOLD, f"msm_{cname}.npz"))
    refc = ref.RefCurve(cname)
    n = g["bases"].shape[0]
    assert np.array_equal(refc.to_affine(refc.msm(np.ascontiguousarray(g["scalars"][:n]), g["bases"])), g["res_single"])
    assert np.array_equal(refc.to_affine(refc.msm(g["scalars"], g["bases"], batch=2)), g["res_batch2_shared"])
    assert np.array_equal(refc.to_affine(refc.msm(g["scalars_mont"], g["bases"], scalars_mont=True)), g["res_mont"])


@pytest.mark.parametrize("cname", CURVES)
def test_python_definition_matches_golden_prefix(cname):
    C = pyref.CURVES[cname]
    g = np.load(os.path.join(GOLD, f"msm_{cname}.npz"))
    L = C.limbs_q
    m = 24
    pts = [(from_words(b[:L]), from_words(b[L:])) for b in g["bases"][:m]]
    assert all(pyref.on_curve(C, p) for p in pts)
    sc = from_words(g["scalars"][:m])
    exp = pyref.msm_naive(C, sc, pts)
    got = oc.to_affine(cname, oc.msm(cname, np.ascontiguousarray(g["scalars"][:m]), np.ascontiguousarray(g["bases"][:m]), c=5))
    assert (from_words(got[:L]), from_words(got[L:])) == exp


@pytest.mark.parametrize("fname", FIELDS)
def test_c_oracle_ntt_matches_golden(fname):
    g = np.load(os.path.join(GOLD, f"ntt_{fname}.npz"))
    root, cg = int(g["domain_root"][0]), int(g["coset_gen"][0])
    x, n, batch = g["x"], 1024, 2
    assert np.array_equal(oc.ntt(fname, x, n, root, batch=batch), g["fwd_NN"])
    assert np.array_equal(oc.ntt(fname, x, n, root, inverse=True, batch=batch), g["inv_NN"])
    assert np.array_equal(oc.ntt(fname, x, n, root, ordering=1, coset_gen=cg, batch=batch), g["fwd_NR_coset"])
    assert np.array_equal(oc.ntt(fname, x, n, root, inverse=True, ordering=2, coset_gen=cg, batch=batch), g["inv_RN_coset"])
    assert np.array_equal(oc.ntt(fname, x, n, root, ordering=3, batch=batch), g["fwd_RR"])
    assert np.array_equal(oc.ntt(fname, x, n, root, batch=batch, columns_batch=True), g["fwd_columns"])
    assert np.array_equal(oc.ntt(fname, g["x_ext"], 64, root, lanes=4), g["fwd_ext"])
    # round trip
    assert np.array_equal(oc.ntt(fname, g["fwd_NN"], n, root, inverse=True, batch=batch), x)


@pytest.mark.parametrize("fname", FIELDS)
def test_reference_ntt_matches_golden_and_definition(fname):
    F = pyref.NTT_FIELDS[fname]
    g = np.load(os.path.join(GOLD, f"ntt_{fname}.npz"))
    root = int(g["domain_root"][0])
    assert root == pyref.omega(F, 12) == oc.omega(fname, 12)
    # O(N^2) definition on one short row, through the C oracle with the same oversized domain
    rng = np.random.default_rng(4)
    x = rng.integers(0, F.p, size=32, dtype=np.uint32)
    for o in ("NN", "NR", "RN", "RR"):
        for inv in (False, True):
            exp = pyref.ntt_naive(F, [int(v) for v in x], pyref.omega(F, 5), inverse=inv, coset_gen=5, ordering=o)
            assert [int(v) for v in oc.ntt(fname, x, 32, root, inverse=inv, ordering=ORD[o], coset_gen=5)] == exp
    if not ref.available(fname):
        pytest.skip("oracle/_ref not built")
    rf = ref.RefNttField(fname)
    rf.init_domain(root)
    try:
        assert np.array_equal(rf.ntt(g["x"], 1024, 0, batch=2), g["fwd_NN"])
        assert np.array_equal(rf.ntt(g["x"], 1024, 1, batch=2, ordering=2, coset_gen=int(g["coset_gen"][0])), g["inv_RN_coset"])
    finally:
        rf.release_domain()


def test_c_oracle_vs_reference_random_msm():
    if not ref.available("bn254"):
        pytest.skip("oracle/_ref not built")
    C = pyref.BN254
    refc = ref.RefCurve("bn254")
    rng = np.random.default_rng(8)
    bases = refc.generate_affine_points(150)  # the reference's own generator (period-100 repetition)
    sc = to_words(rand_scalars(rng, 150, C.r), 8)
    got = oc.to_affine("bn254", oc.msm("bn254", sc, bases, c=8))
    assert np.array_equal(got, refc.to_affine(refc.msm(sc, bases))[0])


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377"])
def test_reference_g2_msm_matches_python_definition(cname):
    """pins pyref's Fq2 / G2 arithmetic (used by the G2 GPU tests) to the reference's G2_ENABLED build"""
    C = pyref.G2_CURVES[cname]
    rc = ref.RefCurve(cname, g2=True)
    L = C.base.limbs_q
    rng = np.random.default_rng(3)
    n = 12
    pts = pyref.g2_gen_points(C, n, k0=424242)
    pts[4] = pyref.INF2
    arr = np.concatenate([to_words([p[0][0] for p in pts], L), to_words([p[0][1] for p in pts], L),
                          to_words([p[1][0] for p in pts], L), to_words([p[1][1] for p in pts], L)], axis=1)
    sc = rand_scalars(rng, n, C.base.r)
    sc[0], sc[1] = 0, C.base.r - 1
    out = rc.msm(to_words(sc, 8), arr)
    assert rc.is_on_curve(out[0])
    aff = rc.to_affine(out)[0]
    got = tuple((from_words(aff[(2 * k) * L:(2 * k + 1) * L]), from_words(aff[(2 * k + 1) * L:(2 * k + 2) * L])) for k in range(2))
    assert got == pyref.g2_msm_naive(C, sc, pts)


@pytest.mark.parametrize("fname", ["bn254", "bls12_381", "bls12_377", "stark252"])
def test_reference_scalar_field_ntt_matches_definition(fname):
    """pins pyref.ntt_naive over the 256-bit scalar fields (used by tests/test_gpu_ntt_scalar.py)"""
    F = pyref.NTT_FIELDS[fname]
    rf = ref.RefScalarNttField(fname)
    root = rf.get_root_of_unity(1 << 8)
    assert root == pyref.omega(F, 8)
    rf.init_domain(root)
    try:
        rng = np.random.default_rng(0)
        n = 16
        vals = rand_scalars(rng, n, F.p)
        x = to_words(vals, 8).reshape(-1)
        y = rf.ntt(x, n, 0)
        assert from_words(y.reshape(n, 8)) == pyref.ntt_naive(F, vals, pyref.omega(F, 4))
        y = rf.ntt(x, n, 1, coset_gen=12345, ordering=3)
        assert from_words(y.reshape(n, 8)) == pyref.ntt_naive(F, vals, pyref.omega(F, 4), inverse=True, coset_gen=12345, ordering="RR")
    finally:
        rf.release_domain()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377"])
def test_reference_build_matches_new_row_goldens(cname):
    """the committed G2 / scalar-field NTT / ECNTT fixtures are what the reference build produces today"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"msm_g2_{cname}.npz"))
    rc = ref.RefCurve(cname, g2=True)
    n = g["bases"].shape[0]
    assert np.array_equal(rc.to_affine(rc.msm(np.ascontiguousarray(g["scalars"][:n]), g["bases"])), g["res_single"])
    s = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"scalar_ntt_{cname}.npz"))
    sf = ref.RefScalarNttField(cname)
    sf.init_domain(from_words(s["domain_root"]))
    try:
        assert np.array_equal(sf.ntt(s["x"], 512, 0, batch=2), s["fwd_NN"])
        assert np.array_equal(sf.ntt(s["x"], 512, 1, batch=2, ordering=2, coset_gen=from_words(s["coset_gen"])), s["inv_RN_coset"])
        r1 = ref.RefCurve(cname)
        L = pyref.CURVES[cname].limbs_q
        assert np.array_equal(r1.to_affine(r1.ecntt(s["ec_points"], 32, 0).reshape(32, 3 * L)), s["ec_fwd_NN_affine"])
    finally:
        sf.release_domain()


def test_reference_test_binaries_host_arithmetic_suite():
    """The reference's own test sources compiled against the GoogleTest stand-in (oracle/build_ref_tests.sh): the part
    that needs no device under test -- CurveSanity (host curve arithmetic of the reference) -- must pass on the CPU.
    Pins the stand-in itself (TYPED_TEST registration, ASSERT_EQ on points, filters) without a GPU."""
    import os
    import subprocess

    import pytest

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "tests", "test_curve_api_bn254")
    if not os.path.exists(exe):
        pytest.skip("reference tests not built (oracle/build_ref_tests.sh needs /root/reference)")
    r = subprocess.run([exe, "--gtest_filter=CurveSanity*"], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "[  PASSED  ] 4 tests." in out, out[-2000:]
    lst = subprocess.run([exe, "--gtest_list_tests"], capture_output=True, text=True, timeout=60).stdout
    for name in ("CurveApiTest.msm", "CurveApiTest.msm_bitsize", "CurveApiTest.msmG2", "CurveApiTest.ecntt", "CurveSanity/1.ScalarMultTest"):
        assert name in lst


def test_reference_goldilocks_matches_golden_and_definition():
    """pins the goldilocks fixture (8-byte elements, quadratic extension transformed component by component) to the
    reference build and to the O(N^2) definition"""
    if not ref.available("goldilocks"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    F = pyref.GOLDILOCKS
    g = np.load(os.path.join(GOLD, "ntt_goldilocks.npz"))
    rf = ref.RefGoldField()
    rf.init_domain(int(g["domain_root"].view("<u8")[0]))
    try:
        cg = int(g["coset_gen"].view("<u8")[0])
        assert np.array_equal(rf.ntt(g["x"], 1024, 0, batch=2), g["fwd_NN"])
        assert np.array_equal(rf.ntt(g["x"], 1024, 1, batch=2, ordering=2, coset_gen=cg), g["inv_RN_coset"])
        assert np.array_equal(rf.ntt(g["x_ext"], 64, 0, extension=True), g["fwd_ext"])
        row = [int(v) for v in g["x"].view("<u8")[:1024]]
        assert [int(v) for v in g["fwd_NN"].view("<u8")[:1024]] == pyref.ntt_naive(F, row, pyref.omega(F, 10))
        assert [int(v) for v in g["inv_RN_coset"].view("<u8")[:1024]] == pyref.ntt_naive(F, row, pyref.omega(F, 10), inverse=True, coset_gen=cg, ordering="RN")
    finally:
        rf.release_domain()
