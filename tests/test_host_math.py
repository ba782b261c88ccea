"""CPU: the exact arithmetic headers the kernels use (bigfield.hpp, ec.hpp, smallfield.hpp), compiled
for the host with the debug bound tracker on, against Python big-int arithmetic. This pins the
29-bit-radix Montgomery field, the XYZZ mixed add with its exceptional cases, and the complete
projective formulas (reference: icicle/include/icicle/curves/projective.h:73-188) without a GPU."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libhost_math.so")
SRC = os.path.join(HERE, "host_math_harness.cpp")
FIELDS = {0: pyref.BN254.q, 1: pyref.BN254.r, 2: pyref.BLS12_381.q, 3: pyref.BLS12_381.r,
          4: pyref.BLS12_377.q, 5: pyref.BLS12_377.r, 6: pyref.STARK252.p, 7: pyref.GOLDILOCKS.p}
NL = {0: 8, 1: 8, 2: 12, 3: 8, 4: 12, 5: 8, 6: 8, 7: 2}


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    deps = [SRC] + [os.path.join(HERE, "..", "icicle_amd", "csrc", f) for f in ("bigfield.hpp", "fq2.hpp", "ec.hpp", "smallfield.hpp", "goldfield.hpp", "field_consts.h", "glv.hpp", "ec_dbl_quad.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-DBIGFIELD_BOUNDS", "-fPIC", "-shared", SRC, "-o", SO])
    return ctypes.CDLL(SO)


def w(x, n):
    return (ctypes.c_uint32 * n)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def iv(a):
    return sum(int(v) << (32 * i) for i, v in enumerate(a))


@pytest.mark.parametrize("f", [0, 1, 2, 3, 4, 5, 6, 7])
def test_field_ops(lib, f):
    p, n = FIELDS[f], NL[f]
    rnd = random.Random(f)
    special = [0, 1, 2, 3, p - 1, p - 2, (1 << 29) - 1, 1 << 29, 1 << 58, p // 2, p // 2 + 1, (1 << 32) - 1, 1 << 32, p - (1 << 32)]
    r32 = 1 << (32 * n)
    for it in range(600):
        a = special[it % len(special)] if it < 2 * len(special) else rnd.randrange(p)
        b = special[(it * 7) % len(special)] if it < len(special) else rnd.randrange(p)
        out = (ctypes.c_uint32 * n)()
        for op, exp in [(0, a * b % p), (1, a * a % p), (2, (a + b) % p), (3, (a - b) % p), (4, (-a) % p),
                        (5, 2 * (a + b) * (a - 4 * b) % p)]:
            assert lib.host_field_op(f, op, w(a, n), w(b, n), out) == 0
            assert iv(out) == exp, (f, op, a, b)
        lib.host_field_op(f, 6, w(a * r32 % p, n), w(0, n), out)
        assert iv(out) == a  # reference-Montgomery -> canonical
        lib.host_field_op(f, 7, w(a, n), w(0, n), out)
        assert iv(out) == a * r32 % p
        lib.host_field_op(f, 8, w(a, n), w(b, n), out)
        assert out[0] == (1 if a == b else 0)
        if it < 40:  # a^(p-2): the exponent's borrow crosses words when p ends in ...00000001 (three of the fields)
            assert lib.host_field_op(f, 9, w(a, n), w(0, n), out) == 0
            assert iv(out) == (pow(a, -1, p) if a else 0), (f, "inv", a)


@pytest.mark.parametrize("ci,c", [(0, pyref.BN254), (1, pyref.BLS12_381), (4, pyref.BLS12_377), (5, pyref.GRUMPKIN)])
def test_ec_ops(lib, ci, c):
    n32 = c.limbs_q
    rnd = random.Random(42 + ci)

    def run(op, pts, aux=None):
        flat = []
        for (x, y) in pts:
            flat += list(w(x, n32)) + list(w(y, n32))
        arr = (ctypes.c_uint32 * max(1, len(flat)))(*flat)
        out = (ctypes.c_uint32 * (3 * n32))()
        auxa = (ctypes.c_uint32 * len(aux))(*aux) if aux is not None else None
        assert lib.host_ec_op(ci, op, arr, len(pts), auxa, out) == 0
        o = list(out)
        X, Y, Z = iv(o[:n32]), iv(o[n32:2 * n32]), iv(o[2 * n32:])
        assert X < c.q and Y < c.q and Z < c.q
        return pyref.proj_to_affine(c, X, Y, Z), (X, Y, Z)

    G = (c.gx, c.gy)
    assert run(3, [G])[0] == G
    base = pyref.gen_points(c, 30, k0=rnd.randrange(c.r))
    for trial in range(24):
        k = rnd.randrange(1, 16)
        pts = [rnd.choice(base) for _ in range(k)]
        neg = [rnd.randrange(2) for _ in range(k)]
        if trial % 3 == 0:  # doubling right at the start of a bucket
            pts[0:0] = [pts[0]] * 2
            neg[0:0] = [neg[0]] * 2
        if trial % 4 == 0:  # cancellation at the end
            pts.append(pts[-1])
            neg.append(1 - neg[-1])
        if trial % 5 == 0:
            pts.insert(1, pyref.INF)
            neg.insert(1, 0)
        if trial == 7:
            pts, neg = [base[0], base[0]], [0, 1]
        if trial == 8:
            pts, neg = [base[0], base[0], base[1]], [0, 1, 0]
        if trial == 9:
            pts, neg = [base[0]] * 9, [0] * 9
        exp = pyref.INF
        for p_, n_ in zip(pts, neg):
            exp = pyref.ec_add(c, exp, pyref.ec_neg(c, p_) if n_ else p_)
        for op in (0, 1):  # XYZZ accumulate / complete projective sum
            got, raw = run(op, pts, neg)
            assert got == exp, (c.name, trial, op)
            if exp == pyref.INF:
                assert raw[2] == 0 and raw[1] != 0  # (0 : y!=0 : 0), never (0,0,0)
    for k in [0, 1, 2, 3, 5, 255, 256, 32767, 32768, 65535, 1 << 20]:
        assert run(2, [base[3]], [k])[0] == pyref.ec_mul(c, k, base[3])
    got, raw = run(2, [pyref.INF], [5])
    assert got == pyref.INF and raw[1] != 0
    for k in [0, 1, 5, 16]:
        assert run(4, [base[2]], [k])[0] == pyref.ec_mul(c, 1 << k, base[2])
    for k in [0, 1, 2, 7, 64, 300]:  # Jacobian doubling chain of the window combine
        assert run(5, [base[2]], [k])[0] == pyref.ec_mul(c, 1 << k, base[2])
    got, raw = run(5, [pyref.INF], [9])
    assert got == pyref.INF and raw[1] != 0
    if ci in (0, 1, 4):  # the ECNTT butterflies' scalar multiplication: GLV split + joint windows + lazily reduced doublings (one lane's arithmetic of ecntt.hip)
        rnd2 = random.Random(900 + ci)
        ks = [0, 1, 2, 15, 16, 17, c.r - 1, c.r - 2, c.r // 2, (1 << 128) - 1, 1 << 128, (1 << 200) + 12345] + [rnd2.randrange(c.r) for _ in range(25)]
        for k in ks:
            assert run(7, [base[4]], [(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)])[0] == pyref.ec_mul(c, k, base[4]), (c.name, hex(k))
        assert run(7, [pyref.INF], [5, 0, 0, 0, 0, 0, 0, 0])[0] == pyref.INF
    for k in [0, 1, 2, 3, 17, 84, 300]:  # msm_precompute_bases' chain: lazily reduced doubling from Z = 1 (bounds asserted by the tracker)
        for b in (base[2], base[5]):
            assert run(6, [b], [k])[0] == pyref.ec_mul(c, 1 << k, b)
    assert run(6, [pyref.INF], [9])[0] == pyref.INF


@pytest.mark.parametrize("ci,c", [(2, pyref.BN254_G2), (3, pyref.BLS12_381_G2), (6, pyref.BLS12_377_G2)])
def test_g2_ec_ops(lib, ci, c):
    """same cases over Fq2 (fq2.hpp): XYZZ accumulation with its exceptional branches, complete add / dbl,
    small multiples; the bound tracker asserts inside the harness on every field operation."""
    n32 = c.base.limbs_q
    q = c.base.q
    rnd = random.Random(142 + ci)

    def run(op, pts, aux=None):
        flat = []
        for (x, y) in pts:
            flat += list(w(x[0], n32)) + list(w(x[1], n32)) + list(w(y[0], n32)) + list(w(y[1], n32))
        arr = (ctypes.c_uint32 * max(1, len(flat)))(*flat)
        out = (ctypes.c_uint32 * (6 * n32))()
        auxa = (ctypes.c_uint32 * len(aux))(*aux) if aux is not None else None
        assert lib.host_ec_op(ci, op, arr, len(pts), auxa, out) == 0
        o = list(out)
        v = [iv(o[i * n32:(i + 1) * n32]) for i in range(6)]
        assert all(t < q for t in v)
        X, Y, Z = (v[0], v[1]), (v[2], v[3]), (v[4], v[5])
        return pyref.g2_proj_to_affine(c, X, Y, Z), (X, Y, Z)

    G = (c.gx, c.gy)
    assert run(3, [G])[0] == G
    base = pyref.g2_gen_points(c, 20, k0=rnd.randrange(c.base.r))
    for trial in range(16):
        k = rnd.randrange(1, 14)
        pts = [rnd.choice(base) for _ in range(k)]
        neg = [rnd.randrange(2) for _ in range(k)]
        if trial % 3 == 0:
            pts[0:0] = [pts[0]] * 2
            neg[0:0] = [neg[0]] * 2
        if trial % 4 == 0:
            pts.append(pts[-1])
            neg.append(1 - neg[-1])
        if trial % 5 == 0:
            pts.insert(1, pyref.INF2)
            neg.insert(1, 0)
        if trial == 7:
            pts, neg = [base[0], base[0]], [0, 1]
        if trial == 8:
            pts, neg = [base[0], base[0], base[1]], [0, 1, 0]
        if trial == 9:
            pts, neg = [base[0]] * 9, [0] * 9
        exp = pyref.INF2
        for p_, n_ in zip(pts, neg):
            exp = pyref.g2_add(c, exp, pyref.g2_neg(c, p_) if n_ else p_)
        for op in (0, 1):
            got, raw = run(op, pts, neg)
            assert got == exp, (c.name, trial, op)
            if exp == pyref.INF2:
                assert raw[2] == (0, 0) and raw[1] != (0, 0)
    for k in [0, 1, 2, 3, 5, 255, 32768, 1 << 20]:
        assert run(2, [base[3]], [k])[0] == pyref.g2_mul(c, k, base[3])
    for k in [0, 1, 5, 16]:
        assert run(4, [base[2]], [k])[0] == pyref.g2_mul(c, 1 << k, base[2])
    for k in [0, 1, 2, 7, 64, 300]:
        assert run(5, [base[2]], [k])[0] == pyref.g2_mul(c, 1 << k, base[2])
    for k in [0, 1, 17, 84]:  # (over Fq2 the precompute chain is dbl_jac itself)
        assert run(6, [base[2]], [k])[0] == pyref.g2_mul(c, 1 << k, base[2])
    got, raw = run(5, [pyref.INF2], [9])
    assert got == pyref.INF2 and raw[1] != (0, 0)


def _consts(cname):
    """GLV constants of struct <cname>_g1 in the generated header"""
    import re

    text = open(os.path.join(HERE, "..", "icicle_amd", "csrc", "field_consts.h")).read()
    body = text[text.index(f"struct {cname}_g1 {{"):]
    body = body[:body.index("\n};")]

    def arr(name):
        m = re.search(name + r"\[\d+\] = \{([^}]*)\}", body)
        return [int(v.rstrip("u"), 16) for v in m.group(1).split(",")]

    return arr


@pytest.mark.parametrize("ci,c", [(0, pyref.BN254), (1, pyref.BLS12_381), (4, pyref.BLS12_377), (5, pyref.GRUMPKIN)])
def test_glv_decomposition(lib, ci, c):
    """glv.hpp (the ECNTT butterflies' scalar split): k = k1 + k2 lambda mod r with both halves below 2^129, for edge values and
    random scalars; lambda acts as phi(x, y) = (beta x, y) on the generator (the constants of tools/gen_consts.py pair up)."""
    arr = _consts(c.name)
    lam = sum(v << (32 * i) for i, v in enumerate(arr("GLV_LAMBDA")))
    r, q = c.r, c.q
    assert (lam * lam + lam + 1) % r == 0
    nl = len(arr("GLV_BETA"))
    beta_m = sum(v << (29 * i) for i, v in enumerate(arr("GLV_BETA")))
    beta = beta_m * pow(1 << (29 * nl), -1, q) % q
    assert pow(beta, 3, q) == 1 and beta != 1
    assert pyref.ec_mul(c, lam, (c.gx, c.gy)) == (beta * c.gx % q, c.gy)
    rnd = random.Random(77 + ci)
    ks = [0, 1, 2, 3, r - 1, r - 2, lam, lam - 1, lam + 1, r // 2, r // 3, (1 << 128) - 1, 1 << 128, (1 << 253) % r] + [rnd.randrange(r) for _ in range(3000)]
    out = (ctypes.c_uint32 * 12)()
    worst = 0
    for k in ks:
        assert lib.host_glv_decompose(ci, w(k, 8), out) == 0
        k1 = sum(int(out[i]) << (32 * i) for i in range(5)) * (-1 if out[5] else 1)
        k2 = sum(int(out[6 + i]) << (32 * i) for i in range(5)) * (-1 if out[11] else 1)
        assert (k1 + k2 * lam - k) % r == 0, (c.name, k)
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    assert worst <= 129, worst


def test_glv_signed_five_bit_recoding(lib):
    """glv.hpp glv_recode5 (the ECNTT chain's windows): 27 digits in [-15, 16] that sum back to k, for every k < 2^130 tried -- the
    all-ones and carry-chain patterns first"""
    rnd = random.Random(55)
    ks = [0, 1, 16, 17, 31, 32, (1 << 130) - 1, (1 << 129) - 1, 1 << 129, (1 << 129) + 1, int("10001" * 26, 2), int("10000" * 26, 2), int("01111" * 26, 2)]
    ks += [rnd.randrange(1 << rnd.randrange(1, 131)) for _ in range(4000)]
    dig = (ctypes.c_int32 * 27)()
    for k in ks:
        assert lib.host_glv_recode5(w(k, 5), dig) == 0
        ds = list(dig)
        assert all(-15 <= d <= 16 for d in ds), (hex(k), ds)
        assert sum(d << (5 * i) for i, d in enumerate(ds)) == k, hex(k)


@pytest.mark.parametrize("fi,f", [(0, pyref.BABYBEAR), (1, pyref.KOALABEAR)])
def test_small_field(lib, fi, f):
    rnd = random.Random(fi)
    out = ctypes.c_uint32()
    for it in range(1500):
        a, b = rnd.randrange(f.p), rnd.randrange(f.p)
        if it < 4:
            a, b = [(0, 0), (f.p - 1, f.p - 1), (1, f.p - 1), (f.p - 1, 0)][it]
        for op, exp in [(0, a * b % f.p), (1, (a + b) % f.p), (2, (a - b) % f.p), (3, pow(a, b, f.p))]:
            lib.host_small_op(fi, op, a, b, ctypes.byref(out))
            assert out.value == exp, (f.name, op, a, b)
        if a:
            lib.host_small_op(fi, 4, a, 0, ctypes.byref(out))
            assert out.value == pow(a, -1, f.p)
