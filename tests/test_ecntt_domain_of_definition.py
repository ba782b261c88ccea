"""Where the ECNTT is defined, and why the endomorphism split of its butterflies (icicle_amd/csrc/glv.hpp) loses nothing.

X[k] = sum_j w^(jk) P_j treats the points as a module over the scalar field Fr: w^a (w^b P) = w^(a + b mod r) P holds when r P = O and
not otherwise. BN254's G1 is the whole curve (cofactor 1). BLS12-381 / BLS12-377 have cofactors of ~2^126, and on a curve point outside
the subgroup of order r the REFERENCE's own transform is not a transform: inverse(forward(x)) != x and forward(x) is not the O(n^2)
definition, because its mixed-radix factorisation composes twiddles that are only equal mod r (backend/cpu/include/ntt_cpu.h:70-232 with
E = projective_t, cpu_ecntt.cpp:13-20). The ECNTT is therefore defined on G1 only -- exactly where phi(P) = lambda P holds and the
split k P = k1 P + k2 phi(P) returns the same group element as the reference's double-and-add (projective.h:192-224). No GPU needed:
this pins the reference's behaviour with the reference (oracle/_ref)."""
import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import from_words, to_words


def _sqrt_mod(a, q):
    """a square root of a mod the prime q, or None (Tonelli-Shanks)"""
    a %= q
    if a == 0:
        return 0
    if pow(a, (q - 1) // 2, q) != 1:
        return None
    if q % 4 == 3:
        return pow(a, (q + 1) // 4, q)
    s, t = 0, q - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (q - 1) // 2, q) != q - 1:
        z += 1
    m, c, u, r = s, pow(z, t, q), pow(a, t, q), pow(a, (t + 1) // 2, q)
    while u != 1:
        i, v = 0, u
        while v != 1:
            v, i = v * v % q, i + 1
        b = pow(c, 1 << (m - i - 1), q)
        m, c, u, r = i, b * b % q, u * b * b % q, r * b % q
    return r


def _curve_points(C, n):
    """(x, sqrt(x^3 + b)) for successive x: spread over E(Fq), in the subgroup of order r with probability 1 / cofactor"""
    pts, x = [], 5
    while len(pts) < n:
        y = _sqrt_mod(x * x * x + C.b, C.q)
        if y is not None:
            pts.append((x, y))
        x += 1
    return pts


def _projective(C, pts):
    L = C.limbs_q
    return np.ascontiguousarray(np.stack([np.concatenate([to_words([p[0]], L)[0], to_words([p[1]], L)[0], to_words([1], L)[0]]) for p in pts]).astype(np.uint32)).reshape(-1)


def _affine(refc, C, words, n):
    L = C.limbs_q
    return [(from_words(a[:L]), from_words(a[L:])) for a in refc.to_affine(words.reshape(n, 3 * L))]


@pytest.mark.parametrize("cname", ["bn254", "bls12_381", "bls12_377"])
def test_reference_ecntt_is_a_transform_on_the_prime_order_subgroup_only(cname):
    C, F = pyref.CURVES[cname], pyref.NTT_FIELDS[cname]
    try:
        refc, sf = ref.RefCurve(cname), ref.RefScalarNttField(cname)
    except Exception as e:  # the reference build for this curve is not in oracle/_ref
        pytest.skip(repr(e))
    n, logn = 8, 3
    root = pyref.omega(F, logn)
    sf.init_domain(root)
    try:
        anywhere = _curve_points(C, n)
        assert all(pyref.on_curve(C, p) for p in anywhere)
        in_subgroup = pyref.ec_mul(C, C.r - 1, anywhere[0]) == pyref.ec_neg(C, anywhere[0])  # r P = O (ec_mul reduces its scalar mod r)
        for pts, is_module in ((anywhere, in_subgroup), ([pyref.ec_mul(C, 3 + 7 * i, (C.gx, C.gy)) for i in range(n)], True)):
            x = _projective(C, pts)
            fwd = refc.ecntt(x, n, 0)
            back = refc.ecntt(fwd, n, 1)
            round_trip = _affine(refc, C, back, n) == pts
            by_definition = _affine(refc, C, fwd, n) == pyref.ecntt_naive(C, F, pts, root)
            assert round_trip == is_module and by_definition == is_module, (cname, is_module, round_trip, by_definition)
        # cofactor 1: every curve point is in G1; the BLS curves: a generic curve point is not
        assert in_subgroup == (cname == "bn254")
    finally:
        sf.release_domain()
