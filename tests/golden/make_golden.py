#!/usr/bin/env python3
"""Mint the golden fixtures in this directory from the REAL reference (oracle/_ref = unmodified ICICLE
CPU backend, built by oracle/build_ref.sh where /root/reference exists).

The reference tree holds no golden vectors / KATs for MSM or NTT (its tests are differential,
SURVEY.md 4, 8c), so these fixtures are what pins the oracle and the HIP path to absolute values:
  python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
Inputs are seeded; inputs AND reference outputs are stored, so the fixtures are usable on the GPU box
where neither /root/reference nor (necessarily) oracle/_ref exists.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyref, ref  # noqa: E402
from tests.util import points_to_array, rand_scalars, to_words  # noqa: E402


def msm_fixture(cname, seed):
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    rng = np.random.default_rng(seed)
    n = 257
    pts = pyref.gen_points(C, n, k0=0xC0FFEE + seed)
    pts[3] = pyref.INF
    pts[100] = pts[101]
    pts[150] = pyref.ec_neg(C, pts[151])
    sc = rand_scalars(rng, 2 * n, C.r)
    sc[0], sc[1], sc[2] = 0, 1, C.r - 1
    sc[100] = sc[101] = 12345
    sc[150] = sc[151] = 77
    scalars = to_words(sc, 8)
    bases = points_to_array(C, pts)
    out = {"scalars": scalars, "bases": bases}
    out["res_single"] = refc.to_affine(refc.msm(scalars[:n], bases))
    out["res_batch2_shared"] = refc.to_affine(refc.msm(scalars, bases, batch=2, shared=True))
    small = to_words(rand_scalars(rng, n, C.r, bits=20), 8)
    out["scalars_20bit"] = small
    out["res_bitsize20"] = refc.to_affine(refc.msm(small, bases, bitsize=20))
    out["scalars_mont"] = refc.scalars_to_montgomery(scalars[:n])
    out["res_mont"] = refc.to_affine(refc.msm(out["scalars_mont"], bases, scalars_mont=True))
    assert np.array_equal(out["res_mont"], out["res_single"])
    # the pure-Python definition agrees on a prefix
    m = 40
    exp = pyref.msm_naive(C, sc[:m], pts[:m])
    got = refc.to_affine(refc.msm(scalars[:m], bases[:m]))
    assert points_to_array(C, [exp]).tolist() == got.tolist()
    return out


def ntt_fixture(fname, seed):
    F = pyref.NTT_FIELDS[fname]
    rf = ref.RefNttField(fname)
    rng = np.random.default_rng(seed)
    logn, batch = 10, 2
    n = 1 << logn
    root = rf.get_root_of_unity(1 << 12)  # domain larger than the transform (stride max/N = 4)
    assert root == pyref.omega(F, 12)
    rf.init_domain(root)
    x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
    g = int(rng.integers(2, F.p))
    out = {"x": x, "domain_root": np.array([root], dtype=np.uint32), "coset_gen": np.array([g], dtype=np.uint32)}
    out["fwd_NN"] = rf.ntt(x, n, 0, batch=batch)
    out["inv_NN"] = rf.ntt(x, n, 1, batch=batch)
    out["fwd_NR_coset"] = rf.ntt(x, n, 0, batch=batch, ordering=1, coset_gen=g)
    out["inv_RN_coset"] = rf.ntt(x, n, 1, batch=batch, ordering=2, coset_gen=g)
    out["fwd_RR"] = rf.ntt(x, n, 0, batch=batch, ordering=3)
    out["fwd_columns"] = rf.ntt(x, n, 0, batch=batch, columns_batch=True)
    xe = rng.integers(0, F.p, size=64 * 4, dtype=np.uint32)
    out["x_ext"] = xe
    out["fwd_ext"] = rf.ntt(xe, 64, 0, extension=True)
    # definition check on one small row
    small = [int(v) for v in x[:16]]
    assert [int(v) for v in rf.ntt(x[:16].copy(), 16, 0)] == pyref.ntt_naive(F, small, pyref.omega(F, 4))
    rf.release_domain()
    return out


def gold_fixture(seed):
    """goldilocks: 8-byte elements, quadratic extension (the reference built with FIELD_ID 1005, EXT_FIELD)"""
    F = pyref.GOLDILOCKS
    rf = ref.RefGoldField()
    rng = np.random.default_rng(seed)
    logn, batch = 10, 2
    n = 1 << logn
    root = rf.get_root_of_unity(1 << 12)
    assert root == pyref.omega(F, 12)
    rf.init_domain(root)

    def elems(count):
        vals = rand_scalars(rng, count, F.p)
        return np.ascontiguousarray(to_words(vals, 2).reshape(-1)), vals

    x, xv = elems(n * batch)
    g = rand_scalars(rng, 1, F.p)[0]
    out = {"x": x, "domain_root": to_words([root], 2)[0], "coset_gen": to_words([g], 2)[0]}
    out["fwd_NN"] = rf.ntt(x, n, 0, batch=batch)
    out["inv_NN"] = rf.ntt(x, n, 1, batch=batch)
    out["fwd_NR_coset"] = rf.ntt(x, n, 0, batch=batch, ordering=1, coset_gen=g)
    out["inv_RN_coset"] = rf.ntt(x, n, 1, batch=batch, ordering=2, coset_gen=g)
    out["fwd_RR"] = rf.ntt(x, n, 0, batch=batch, ordering=3)
    out["fwd_columns"] = rf.ntt(x, n, 0, batch=batch, columns_batch=True)
    xe, xev = elems(64 * 2)
    out["x_ext"] = xe
    out["fwd_ext"] = rf.ntt(xe, 64, 0, extension=True)
    # the definition: one small row, and the extension transform component by component
    small, sv = elems(16)
    assert [int(v) for v in rf.ntt(small, 16, 0).view("<u8")] == pyref.ntt_naive(F, sv, pyref.omega(F, 4))
    for k in range(2):
        comp = pyref.ntt_naive(F, xev[k::2], pyref.omega(F, 6))
        assert [int(v) for v in out["fwd_ext"].view("<u8")[k::2]] == comp
    rf.release_domain()
    return out


def g2_fixture(cname, seed):
    """G2 MSM (reference built with G2_ENABLED): inputs, affine results"""
    C = pyref.G2_CURVES[cname]
    refc = ref.RefCurve(cname, g2=True)
    L = C.base.limbs_q
    rng = np.random.default_rng(seed)
    n = 129
    pts = pyref.g2_gen_points(C, n, k0=0xBEEF + seed)
    pts[5] = pyref.INF2
    pts[40] = pts[41]
    pts[60] = pyref.g2_neg(C, pts[61])
    sc = rand_scalars(rng, 2 * n, C.base.r)
    sc[0], sc[1], sc[2] = 0, 1, C.base.r - 1
    sc[40] = sc[41] = 999
    sc[60] = sc[61] = 31
    bases = np.concatenate([to_words([p[0][0] for p in pts], L), to_words([p[0][1] for p in pts], L),
                            to_words([p[1][0] for p in pts], L), to_words([p[1][1] for p in pts], L)], axis=1)
    scalars = to_words(sc, 8)
    out = {"scalars": scalars, "bases": bases}
    out["res_single"] = refc.to_affine(refc.msm(scalars[:n], bases))
    out["res_batch2_shared"] = refc.to_affine(refc.msm(scalars, bases, batch=2, shared=True))
    exp = pyref.g2_msm_naive(C, sc[:24], pts[:24])
    got = refc.to_affine(refc.msm(scalars[:24], bases[:24]))[0]
    assert [int(v) for v in got] == [int(v) for v in np.concatenate([to_words([exp[0][0]], L)[0], to_words([exp[0][1]], L)[0],
                                                                     to_words([exp[1][0]], L)[0], to_words([exp[1][1]], L)[0]])]
    return out


def scalar_ntt_fixture(cname, seed, with_ec=True):
    """NTT over the curve's 256-bit scalar field + ECNTT over G1 on the same domain (with_ec=False: a 256-bit field
    without a curve, stark252)"""
    F = pyref.NTT_FIELDS[cname]
    sf = ref.RefScalarNttField(cname)
    rng = np.random.default_rng(seed)
    logn, batch = 9, 2
    n = 1 << logn
    root = sf.get_root_of_unity(1 << 11)
    assert root == pyref.omega(F, 11)
    sf.init_domain(root)
    x = to_words(rand_scalars(rng, n * batch, F.p), 8).reshape(-1)
    g = rand_scalars(rng, 1, F.p)[0]
    out = {"x": x, "domain_root": to_words([root], 8)[0], "coset_gen": to_words([g], 8)[0]}
    out["fwd_NN"] = sf.ntt(x, n, 0, batch=batch)
    out["inv_NN"] = sf.ntt(x, n, 1, batch=batch)
    out["fwd_NR_coset"] = sf.ntt(x, n, 0, batch=batch, ordering=1, coset_gen=g)
    out["inv_RN_coset"] = sf.ntt(x, n, 1, batch=batch, ordering=2, coset_gen=g)
    out["fwd_columns"] = sf.ntt(x, n, 0, batch=batch, columns_batch=True)
    small = [sum(int(v) << (32 * k) for k, v in enumerate(x[8 * i:8 * i + 8])) for i in range(16)]
    assert [sum(int(v) << (32 * k) for k, v in enumerate(r)) for r in sf.ntt(x[:128].copy(), 16, 0).reshape(16, 8)] == pyref.ntt_naive(F, small, pyref.omega(F, 4))
    if not with_ec:
        sf.release_domain()
        return out
    C = pyref.CURVES[cname]
    refc = ref.RefCurve(cname)
    m = 32
    pts = pyref.gen_points(C, m, k0=4242 + seed)
    pts[7] = pyref.INF
    Lq = C.limbs_q
    rows = []
    for p_ in pts:
        xyz = (0, 1, 0) if p_ == pyref.INF else (p_[0], p_[1], 1)
        rows.append(np.concatenate([to_words([v], Lq)[0] for v in xyz]))
    proj = np.ascontiguousarray(np.stack(rows).astype(np.uint32)).reshape(-1)
    out["ec_points"] = proj
    out["ec_fwd_NN_affine"] = refc.to_affine(refc.ecntt(proj, m, 0).reshape(m, 3 * Lq))
    out["ec_inv_NR_coset_affine"] = refc.to_affine(refc.ecntt(proj, m, 1, ordering=1, coset_gen=g).reshape(m, 3 * Lq))
    assert [(int(sum(int(v) << (32 * k) for k, v in enumerate(a[:Lq]))), int(sum(int(v) << (32 * k) for k, v in enumerate(a[Lq:]))))
            for a in out["ec_fwd_NN_affine"][:4]] == pyref.ecntt_naive(C, F, pts, pyref.omega(F, 5))[:4]
    sf.release_domain()
    return out


def main():
    """`--missing`: mint only the fixtures that are not there yet (inputs are seeded: the others would come out the same)"""
    missing_only = "--missing" in sys.argv

    def save(name, make):
        path = os.path.join(HERE, name)
        if missing_only and os.path.exists(path):
            return
        np.savez_compressed(path, **make())
        print("wrote", name)

    for i, c in enumerate(("bn254", "bls12_381", "bls12_377", "grumpkin")):
        save(f"msm_{c}.npz", lambda: msm_fixture(c, 11 + i))
    for i, f in enumerate(("babybear", "koalabear")):
        save(f"ntt_{f}.npz", lambda: ntt_fixture(f, 21 + i))
    for i, c in enumerate(("bn254", "bls12_381", "bls12_377")):
        save(f"msm_g2_{c}.npz", lambda: g2_fixture(c, 31 + i))
        save(f"scalar_ntt_{c}.npz", lambda: scalar_ntt_fixture(c, 41 + i))
    save("scalar_ntt_stark252.npz", lambda: scalar_ntt_fixture("stark252", 44, with_ec=False))
    save("ntt_goldilocks.npz", lambda: gold_fixture(51))
    print("golden fixtures in", HERE)


if __name__ == "__main__":
    main()
