"""TEST INFRASTRUCTURE ONLY (never imported by icicle_amd/).

Pure-Python big-integer restatement of the definitions the MSM / NTT hot path must satisfy, used
to pin both the C oracle (oracle/oracle.c) and the reference build (oracle/_ref) on small cases:

  * MSM:  result = sum_i s_i * P_i on y^2 = x^3 + b over F_q, affine identity = (0,0)
          (reference: icicle/include/icicle/curves/affine.h:16,28; projective.h:55-59 to_affine;
           icicle/backend/cpu/src/curve/cpu_msm.hpp:431-443 for the batch/precompute layout).
  * NTT:  X[k] = sum_j x[j] w^(jk), inverse scaled by N^-1, coset / ordering / batch layouts as in
          icicle/backend/cpu/include/ntt_cpu.h:70-232, 247-306.

Curve and field parameters are the published ones; tests/test_consts.py cross-checks them against
the reference headers when /root/reference is available.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Curve:
    name: str
    q: int  # base field modulus
    r: int  # scalar field modulus (group order)
    b: int
    gx: int
    gy: int
    limbs_q: int  # 32-bit limbs per base-field element
    limbs_r: int


BN254 = Curve(
    "bn254",
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    3,
    1,
    2,
    8,
    8,
)
BLS12_381 = Curve(
    "bls12_381",
    0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    4,
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    12,
    8,
)
# reference: curves/params/bls12_377.h, fields/snark_fields/bls12_377_{base,scalar}.h
BLS12_377 = Curve(
    "bls12_377",
    0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
    0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
    1,
    0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
    0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
    12,
    8,
)
# reference: curves/params/grumpkin.h (y^2 = x^3 - 17 over BN254's scalar field; group order = BN254's base modulus)
GRUMPKIN = Curve(
    "grumpkin",
    BN254.r,
    BN254.q,
    BN254.r - 17,
    1,
    0x0000000000000002CF135E7506A45D632D270D45F1181294833FC48D823F272C,
    8,
    8,
)
CURVES = {"bn254": BN254, "bls12_381": BLS12_381, "bls12_377": BLS12_377, "grumpkin": GRUMPKIN}

INF = (0, 0)  # the reference's affine identity encoding


def ec_add(c: Curve, p, q):
    if p == INF:
        return q
    if q == INF:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % c.q == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, c.q) % c.q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, c.q) % c.q
    x3 = (lam * lam - x1 - x2) % c.q
    return (x3, (lam * (x1 - x3) - y1) % c.q)


def ec_neg(c: Curve, p):
    return INF if p == INF else (p[0], (-p[1]) % c.q)


def ec_mul(c: Curve, k: int, p):
    k %= c.r
    acc = INF
    while k:
        if k & 1:
            acc = ec_add(c, acc, p)
        p = ec_add(c, p, p)
        k >>= 1
    return acc


def on_curve(c: Curve, p):
    return p == INF or (p[1] * p[1] - p[0] ** 3 - c.b) % c.q == 0


def msm_naive(c: Curve, scalars, points):
    acc = INF
    for s, p in zip(scalars, points):
        acc = ec_add(c, acc, ec_mul(c, s, p))
    return acc


def proj_to_affine(c: Curve, x, y, z):
    """Projective{x,y,z} -> affine, with the reference's convention inverse(0) = 0 => (0,0)."""
    if z % c.q == 0:
        return INF
    zi = pow(z, -1, c.q)
    return (x * zi % c.q, y * zi % c.q)


def gen_points(c: Curve, n: int, k0: int = 1):
    """n distinct affine points (k0 + i) * G, by a running affine add."""
    g = (c.gx, c.gy)
    p = ec_mul(c, k0, g)
    out = []
    for _ in range(n):
        out.append(p)
        p = ec_add(c, p, g)
    return out


# ----------------------------------------------------------------------------------------------
# G2: the sextic twists over Fq2 = Fq[u]/(u^2 + nonresidue) (reference: ComplexExtensionField, c0 = real,
# c1 = imaginary; curves/params/bn254.h:32-53, bls12_381.h / bls12_377.h G2 blocks; nonresidue 1 except BLS12-377's 5,
# fields/snark_fields/bls12_377_base.h:1560-1564). Elements are (c0, c1) tuples; points ((x0,x1),(y0,y1)).
_NONRES = {BLS12_377.q: 5}


def f2_add(q, a, b):
    return ((a[0] + b[0]) % q, (a[1] + b[1]) % q)


def f2_sub(q, a, b):
    return ((a[0] - b[0]) % q, (a[1] - b[1]) % q)


def f2_mul(q, a, b):
    return ((a[0] * b[0] - _NONRES.get(q, 1) * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def f2_inv(q, a):
    ni = pow((a[0] * a[0] + _NONRES.get(q, 1) * a[1] * a[1]) % q, -1, q)
    return (a[0] * ni % q, (-a[1]) * ni % q)


@dataclass(frozen=True)
class G2Curve:
    name: str
    base: Curve
    b: tuple
    gx: tuple
    gy: tuple


BN254_G2 = G2Curve(
    "bn254",
    BN254,
    f2_mul(BN254.q, (3, 0), f2_inv(BN254.q, (9, 1))),
    (
        10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634,
    ),
    (
        8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531,
    ),
)
BLS12_381_G2 = G2Curve(
    "bls12_381",
    BLS12_381,
    (4, 4),
    (
        0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
    ),
    (
        0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
    ),
)
BLS12_377_G2 = G2Curve(
    "bls12_377",
    BLS12_377,
    (0, 0x010222F6DB0FD6F343BD03737460C589DC7B4F91CD5FD889129207B63C6BF8000DD39E5C1CCCCCCD1C9ED9999999999A),
    (
        0x018480BE71C785FEC89630A2A3841D01C565F071203E50317EA501F557DB6B9B71889F52BB53540274E3E48F7C005196,
        0x00EA6040E700403170DC5A51B1B140D5532777EE6651CECBE7223ECE0799C9DE5CF89984BFF76FE6B26BFEFA6EA16AFE,
    ),
    (
        0x00690D665D446F7BD960736BCBB2EFB4DE03ED7274B49A58E458C282F832D204F2CF88886D8C7C2EF094094409FD4DDF,
        0x00F8169FD28355189E549DA3151A70AA61EF11AC3D591BF12463B01ACEE304C24279B83F5E52270BD9A1CDD185EB8F93,
    ),
)
G2_CURVES = {"bn254": BN254_G2, "bls12_381": BLS12_381_G2, "bls12_377": BLS12_377_G2}
INF2 = ((0, 0), (0, 0))


def g2_add(c: G2Curve, p, r):
    q = c.base.q
    if p == INF2:
        return r
    if r == INF2:
        return p
    (x1, y1), (x2, y2) = p, r
    if x1 == x2:
        if f2_add(q, y1, y2) == (0, 0):
            return INF2
        lam = f2_mul(q, f2_mul(q, (3, 0), f2_mul(q, x1, x1)), f2_inv(q, f2_add(q, y1, y1)))
    else:
        lam = f2_mul(q, f2_sub(q, y2, y1), f2_inv(q, f2_sub(q, x2, x1)))
    x3 = f2_sub(q, f2_sub(q, f2_mul(q, lam, lam), x1), x2)
    return (x3, f2_sub(q, f2_mul(q, lam, f2_sub(q, x1, x3)), y1))


def g2_neg(c: G2Curve, p):
    return INF2 if p == INF2 else (p[0], f2_sub(c.base.q, (0, 0), p[1]))


def g2_mul(c: G2Curve, k: int, p):
    k %= c.base.r
    acc = INF2
    while k:
        if k & 1:
            acc = g2_add(c, acc, p)
        p = g2_add(c, p, p)
        k >>= 1
    return acc


def g2_on_curve(c: G2Curve, p):
    q = c.base.q
    if p == INF2:
        return True
    x, y = p
    return f2_sub(q, f2_mul(q, y, y), f2_add(q, f2_mul(q, f2_mul(q, x, x), x), c.b)) == (0, 0)


def g2_msm_naive(c: G2Curve, scalars, points):
    acc = INF2
    for s, p in zip(scalars, points):
        acc = g2_add(c, acc, g2_mul(c, s, p))
    return acc


def g2_proj_to_affine(c: G2Curve, x, y, z):
    q = c.base.q
    if z == (0, 0):
        return INF2
    zi = f2_inv(q, z)
    return (f2_mul(q, x, zi), f2_mul(q, y, zi))


def g2_gen_points(c: G2Curve, n: int, k0: int = 1):
    g = (c.gx, c.gy)
    p = g2_mul(c, k0, g)
    out = []
    for _ in range(n):
        out.append(p)
        p = g2_add(c, p, g)
    return out


# ----------------------------------------------------------------------------------------------
# small NTT fields
@dataclass(frozen=True)
class NttField:
    name: str
    p: int
    rou: int  # root of unity of order 2^two_adicity (reference fp_config::rou)
    two_adicity: int


BABYBEAR = NttField("babybear", 0x78000001, 0x89, 27)
KOALABEAR = NttField("koalabear", 0x7F000001, 0x6AC49F88, 24)
BN254_FR = NttField(
    "bn254", BN254.r, 0x2A3C09F0A58A7E8500E0A7EB8EF62ABC402D111E41112ED49BD61B6E725B19F0, 28
)
BLS12_381_FR = NttField(
    "bls12_381", BLS12_381.r, 0x0212D79E5B416B6F0FD56DC8D168D6C0C4024FF270B3E0941B788F500B912F1F, 32
)
BLS12_377_FR = NttField(
    "bls12_377", BLS12_377.r, 0x11D4B7F60CB92CC160C69477D1A8A12F9B506EE363E3F04A476EF4A4EC2A895E, 47
)
STARK252 = NttField(
    "stark252",
    0x0800000000000011000000000000000000000000000000000000000000000001,
    0x005282DB87529CFA3F0464519C8B0FA5AD187148E11A61616070024F42F8EF94,
    192,
)
# goldilocks, 2^64 - 2^32 + 1 (fields/stark_fields/goldilocks.h:206-276); quadratic extension u^2 = 7, NTT lane-wise
GOLDILOCKS = NttField("goldilocks", 0xFFFFFFFF00000001, 0x185629DCDA58878C, 32)
NTT_FIELDS = {
    "goldilocks": GOLDILOCKS,
    "babybear": BABYBEAR,
    "koalabear": KOALABEAR,
    "bn254": BN254_FR,
    "bls12_381": BLS12_381_FR,
    "bls12_377": BLS12_377_FR,
    "stark252": STARK252,
}


def omega(f: NttField, logn: int) -> int:
    """ModArith::omega (modular_arithmetic.h:61-73): rou^(2^(two_adicity-logn))."""
    if logn == 0:
        return 1
    assert logn <= f.two_adicity
    return pow(f.rou, 1 << (f.two_adicity - logn), f.p)


def bitrev(i: int, logn: int) -> int:
    r = 0
    for _ in range(logn):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def ntt_naive(f: NttField, x, w_n: int, inverse: bool = False, coset_gen: int = 1, ordering: str = "NN"):
    """O(N^2) definition of the reference transform for ONE row:
    ordering[0]=='R': memory index bitrev(j) holds logical x[j]; ordering[1]=='R': logical X[k]
    stored at bitrev(k). 'M' behaves like 'N' (what the CPU backend does).
    forward: x[j] *= g^j first; inverse: result[j] *= g^-j * N^-1 after."""
    n = len(x)
    logn = n.bit_length() - 1
    p = f.p
    xin = list(x)
    if ordering[0] == "R":
        xin = [x[bitrev(j, logn)] for j in range(n)]
    if not inverse and coset_gen != 1:
        xin = [v * pow(coset_gen, j, p) % p for j, v in enumerate(xin)]
    w = pow(w_n, -1, p) if inverse else w_n
    out = [sum(xin[j] * pow(w, j * k, p) for j in range(n)) % p for k in range(n)]
    if inverse:
        ninv = pow(n, -1, p)
        ginv = pow(coset_gen, -1, p)
        out = [v * ninv * pow(ginv, j, p) % p for j, v in enumerate(out)]
    if ordering[1] == "R":
        o2 = [0] * n
        for k in range(n):
            o2[bitrev(k, logn)] = out[k]
        out = o2
    return out


def ecntt_naive(c: Curve, f: NttField, pts, w_n: int, inverse: bool = False, coset_gen: int = 1, ordering: str = "NN"):
    """O(N^2) definition of the reference ECNTT for one row: ntt_naive with group elements in place of field
    elements (the CPU backend instantiates its NTT with E = projective_t, backend/cpu/src/curve/cpu_ecntt.cpp)."""
    n = len(pts)
    logn = n.bit_length() - 1
    p = f.p
    xin = list(pts)
    if ordering[0] == "R":
        xin = [pts[bitrev(j, logn)] for j in range(n)]
    if not inverse and coset_gen != 1:
        xin = [ec_mul(c, pow(coset_gen, j, p), v) for j, v in enumerate(xin)]
    w = pow(w_n, -1, p) if inverse else w_n
    out = []
    for k in range(n):
        acc = INF
        for j in range(n):
            acc = ec_add(c, acc, ec_mul(c, pow(w, j * k, p), xin[j]))
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, p)
        ginv = pow(coset_gen, -1, p)
        out = [ec_mul(c, ninv * pow(ginv, j, p) % p, v) for j, v in enumerate(out)]
    if ordering[1] == "R":
        o2 = [None] * n
        for k in range(n):
            o2[bitrev(k, logn)] = out[k]
        out = o2
    return out
