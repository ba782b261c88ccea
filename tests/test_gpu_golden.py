"""GPU parity against the committed golden fixtures (tests/golden/*.npz, minted from the reference CPU
backend): bit-exact affine results / memcmp-exact NTT outputs. Needs no oracle build on the GPU box."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _affine_py(cname, proj):
    from oracle import pyref
    from tests.util import points_to_array, proj_to_affine_py

    C = pyref.CURVES[cname]
    return np.concatenate([points_to_array(C, [proj_to_affine_py(C, row)[0]]) for row in proj], axis=0)


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377", "grumpkin"])
def test_msm_golden(hip, cname):
    from icicle_amd import msm as M

    g = np.load(os.path.join(GOLD, f"msm_{cname}.npz"))
    n = g["bases"].shape[0]
    cfg = hip.MSMConfig.default()
    assert np.array_equal(_affine_py(cname, M.msm(cname, np.ascontiguousarray(g["scalars"][:n]), g["bases"], cfg)), g["res_single"])
    cfg = hip.MSMConfig.default()
    cfg.batch_size = 2
    assert np.array_equal(_affine_py(cname, M.msm(cname, g["scalars"], g["bases"], cfg)), g["res_batch2_shared"])
    cfg = hip.MSMConfig.default()
    cfg.bitsize = 20
    assert np.array_equal(_affine_py(cname, M.msm(cname, g["scalars_20bit"], g["bases"], cfg)), g["res_bitsize20"])
    cfg = hip.MSMConfig.default()
    cfg.are_scalars_montgomery_form = True
    assert np.array_equal(_affine_py(cname, M.msm(cname, g["scalars_mont"], g["bases"], cfg)), g["res_mont"])


@pytest.mark.parametrize("fname", ["babybear", "koalabear"])
def test_ntt_golden(hip, fname):
    from icicle_amd import ntt as N

    g = np.load(os.path.join(GOLD, f"ntt_{fname}.npz"))
    N.init_domain(fname, int(g["domain_root"][0]))
    try:
        cg = int(g["coset_gen"][0])

        def run(x, direction, **kw):
            cfg = hip.NTTConfigU32.default()
            cfg.batch_size = kw.get("batch", 2)
            cfg.ordering = kw.get("ordering", 0)
            cfg.coset_gen = kw.get("coset", 1)
            cfg.columns_batch = kw.get("columns", False)
            return N.ntt(fname, x, direction, cfg, size=kw.get("size", 1024), extension=kw.get("ext", False))

        assert np.array_equal(run(g["x"], 0), g["fwd_NN"])
        assert np.array_equal(run(g["x"], 1), g["inv_NN"])
        assert np.array_equal(run(g["x"], 0, ordering=1, coset=cg), g["fwd_NR_coset"])
        assert np.array_equal(run(g["x"], 1, ordering=2, coset=cg), g["inv_RN_coset"])
        assert np.array_equal(run(g["x"], 0, ordering=3), g["fwd_RR"])
        assert np.array_equal(run(g["x"], 0, columns=True), g["fwd_columns"])
        assert np.array_equal(run(g["x_ext"], 0, batch=1, size=64, ext=True), g["fwd_ext"])
    finally:
        N.release_domain(fname)


def _g2_affine_py(cname, proj):
    from oracle import pyref
    from tests.util import to_words

    C = pyref.G2_CURVES[cname]
    L = C.base.limbs_q
    rows = []
    for row in proj:
        v = [sum(int(x) << (32 * k) for k, x in enumerate(row[i * L:(i + 1) * L])) for i in range(6)]
        a = pyref.g2_proj_to_affine(C, (v[0], v[1]), (v[2], v[3]), (v[4], v[5]))
        rows.append(np.concatenate([to_words([a[0][0]], L)[0], to_words([a[0][1]], L)[0], to_words([a[1][0]], L)[0], to_words([a[1][1]], L)[0]]))
    return np.stack(rows)


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377"])
def test_g2_msm_golden(hip, cname):
    from icicle_amd import msm as M

    g = np.load(os.path.join(GOLD, f"msm_g2_{cname}.npz"))
    n = g["bases"].shape[0]
    got = M.msm(cname, np.ascontiguousarray(g["scalars"][:n]), g["bases"], g2=True)
    assert np.array_equal(_g2_affine_py(cname, got), g["res_single"])
    cfg = hip.MSMConfig.default()
    cfg.batch_size = 2
    assert np.array_equal(_g2_affine_py(cname, M.msm(cname, g["scalars"], g["bases"], cfg, g2=True)), g["res_batch2_shared"])


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377", "stark252"])
def test_scalar_ntt_and_ecntt_golden(hip, cname):
    from icicle_amd import ntt as N
    from oracle import pyref

    g = np.load(os.path.join(GOLD, f"scalar_ntt_{cname}.npz"))
    words = lambda w: sum(int(x) << (32 * k) for k, x in enumerate(w))
    N.init_domain(cname, words(g["domain_root"]))
    try:
        cg = words(g["coset_gen"])

        def run(direction, **kw):
            cfg = hip.NTTConfigU256.default()
            cfg.batch_size = 2
            cfg.ordering = kw.get("ordering", 0)
            cfg.columns_batch = kw.get("columns", False)
            cfg.set_coset_gen(kw.get("coset", 1))
            return N.ntt(cname, g["x"], direction, cfg, size=512)

        assert np.array_equal(run(0), g["fwd_NN"])
        assert np.array_equal(run(1), g["inv_NN"])
        assert np.array_equal(run(0, ordering=1, coset=cg), g["fwd_NR_coset"])
        assert np.array_equal(run(1, ordering=2, coset=cg), g["inv_RN_coset"])
        assert np.array_equal(run(0, columns=True), g["fwd_columns"])
        if "ec_points" not in g:  # stark252: a field with an NTT and no curve
            return
        m = 32
        y = N.ecntt(cname, g["ec_points"], N.FORWARD, size=m)
        assert np.array_equal(_affine_py(cname, y.reshape(m, -1)), g["ec_fwd_NN_affine"])
        cfg = hip.NTTConfigU256.default()
        cfg.ordering = 1
        cfg.set_coset_gen(cg)
        y = N.ecntt(cname, g["ec_points"], N.INVERSE, cfg, size=m)
        assert np.array_equal(_affine_py(cname, y.reshape(m, -1)), g["ec_inv_NR_coset_affine"])
    finally:
        N.release_domain(cname)
