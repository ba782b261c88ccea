#!/usr/bin/env python3
"""Generate icicle_amd/csrc/field_consts.h: per-field constants for the 29-bit-radix Montgomery
big-field arithmetic (bigfield.hpp) and the 31-bit NTT fields.

Moduli / generators / roots of unity are the published curve parameters; they are cross-checked
against the reference headers (icicle/include/icicle/fields/snark_fields/*.h, stark_fields/*.h,
curves/params/*.h) by tests/test_consts.py when /root/reference is present.
"""
import sys

BIG = {
    # name: (modulus, limbs32)
    "bn254_fq": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 8),
    "bn254_fr": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 8),
    "bls12_381_fq": (
        0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        12,
    ),
    "bls12_381_fr": (0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 8),
    "bls12_377_fq": (
        0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
        12,
    ),
    "bls12_377_fr": (0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001, 8),
    # the Stark curve's base field, 2^251 + 17 * 2^192 + 1: a field with an NTT only (reference FIELD_ID 1002)
    "stark252_fr": (0x0800000000000011000000000000000000000000000000000000000000000001, 8),
}
# quadratic non-residue of the G2 extension Fq2 = Fq[u]/(u^2 + NONRES) (reference fq_config::nonresidue with
# nonresidue_is_negative = true: bn254_base.h:67-71, bls12_381_base.h, bls12_377_base.h:1560-1564)
NONRES = {"bn254_fq": 1, "bls12_381_fq": 1, "bls12_377_fq": 5}
# scalar fields with an NTT: root of unity of order 2^two_adicity (reference fp_config::rou,
# fields/snark_fields/bn254_scalar.h:68-69, bls12_381_scalar.h:46-47)
BIG_ROU = {
    "bn254_fr": 0x2A3C09F0A58A7E8500E0A7EB8EF62ABC402D111E41112ED49BD61B6E725B19F0,
    "bls12_381_fr": 0x0212D79E5B416B6F0FD56DC8D168D6C0C4024FF270B3E0941B788F500B912F1F,
    "bls12_377_fr": 0x11D4B7F60CB92CC160C69477D1A8A12F9B506EE363E3F04A476EF4A4EC2A895E,
    "stark252_fr": 0x005282DB87529CFA3F0464519C8B0FA5AD187148E11A61616070024F42F8EF94,
}
RB = 29


def limbs(x, n, rb=RB, top_unbounded=True):
    out = []
    for i in range(n):
        if i == n - 1 and top_unbounded:
            out.append(x)
        else:
            out.append(x & ((1 << rb) - 1))
            x >>= rb
    assert out[-1] < (1 << 32)
    return out


def arr(v):
    return "{" + ", ".join("0x%08xu" % x for x in v) + "}"


def gen_big(name, p, l32):
    bits = p.bit_length()
    nl = (bits + 6 + RB - 1) // RB  # >= 7 bits of slack: values up to 64p never overflow, R/p >= 2^7... checked below
    R = 1 << (RB * nl)
    assert R // p >= 64, (name, R // p)  # lazy-reduction slack, see bigfield.hpp bounds
    pinv = (-pow(p, -1, 1 << RB)) % (1 << RB)
    r32 = 1 << (32 * l32)
    s = []
    s.append(f"struct {name}_params {{")
    s.append(f"  static constexpr int NL = {nl};      // {RB}-bit limbs")
    s.append(f"  static constexpr int NL32 = {l32};   // packed 32-bit limbs (reference storage<{l32}>)")
    s.append(f"  static constexpr int NBITS = {bits};")
    s.append(f"  static constexpr uint32_t PINV = 0x{pinv:08x}u; // -p^-1 mod 2^{RB}")
    s.append(f"  static constexpr uint32_t P[{nl}] = {arr(limbs(p, nl))};")
    s.append(f"  static constexpr uint32_t P32[{l32}] = {arr(limbs(p, l32, 32))};")
    for k in (2, 4, 8, 16):
        s.append(f"  static constexpr uint32_t P{k}[{nl}] = {arr(limbs(k * p, nl))}; // {k}p")
    s.append(f"  static constexpr uint32_t ONE[{nl}] = {arr(limbs(R % p, nl))}; // R mod p")
    s.append(f"  static constexpr uint32_t R2[{nl}] = {arr(limbs(R * R % p, nl))}; // R^2 mod p")
    # reference-Montgomery (x * 2^(32*l32)) -> canonical, via montmul(x, C): C = R / 2^(32 l32)
    c1 = R * pow(r32, -1, p) % p
    s.append(f"  static constexpr uint32_t REFMONT_TO_CANON[{nl}] = {arr(limbs(c1, nl))}; // R/2^{32*l32}")
    c2 = R * R * pow(r32, -1, p) % p
    s.append(f"  static constexpr uint32_t REFMONT_TO_MONT[{nl}] = {arr(limbs(c2, nl))}; // R^2/2^{32*l32}")
    c4 = r32 * R % p  # canonical x -> reference-Montgomery x*2^(32 l32): montmul(x, C) = x*C/R
    s.append(f"  static constexpr uint32_t CANON_TO_REFMONT[{nl}] = {arr(limbs(c4, nl))}; // 2^{32*l32} * R mod p")
    c3 = r32 % p  # our-Montgomery x*R -> reference-Montgomery: montmul(xR, C) = x*C => C = 2^(32 l32)
    s.append(f"  static constexpr uint32_t MONT_TO_REFMONT[{nl}] = {arr(limbs(c3, nl))}; // 2^{32*l32} mod p")
    if name in NONRES:
        s.append(f"  static constexpr int NONRES = {NONRES[name]}; // Fq2 = Fq[u]/(u^2 + NONRES)")
    if name in BIG_ROU:
        rou = BIG_ROU[name]
        ta = ((p - 1) & -(p - 1)).bit_length() - 1
        assert pow(rou, 1 << ta, p) == 1 and pow(rou, 1 << (ta - 1), p) != 1
        s.append(f"  static constexpr int TWO_ADICITY = {ta};")
        s.append(f"  static constexpr uint32_t ROU32[{l32}] = {arr(limbs(rou, l32, 32))}; // canonical, order 2^TWO_ADICITY")
    s.append("};")
    return "\n".join(s)


CURVES = {
    # name: (base field, scalar field, weierstrass b, generator x, generator y)
    "bn254": ("bn254_fq", "bn254_fr", 3, 1, 2),
    "bls12_381": (
        "bls12_381_fq",
        "bls12_381_fr",
        4,
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    ),
    "bls12_377": (
        "bls12_377_fq",
        "bls12_377_fr",
        1,
        0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
    ),
    # y^2 = x^3 - 17 over BN254's scalar field; its group order is BN254's base-field modulus (the curves form a cycle)
    "grumpkin": ("bn254_fr", "bn254_fq", -17, 1, 0x0000000000000002CF135E7506A45D632D270D45F1181294833FC48D823F272C),
}


# G2: twist curves over Fq2 = Fq[u]/(u^2+1); name: (b' as (c0, c1) or callable, generator x (c0, c1), y (c0, c1))
# generators: the standard ones (EIP-197 for BN254, the BLS12-381 spec), the same values as the reference's
# curves/params/{bn254,bls12_381}.h g2_gen_*; checked on-curve below.
G2 = {
    "bn254": (
        "3/(9+u)",
        (
            10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
        ),
        (
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531,
        ),
    ),
    "bls12_381": (
        "4(1+u)",
        (
            0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
        ),
        (
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
        ),
    ),
    "bls12_377": (
        "(0, 0x010222f6...) over u^2 = -5",
        (
            0x018480BE71C785FEC89630A2A3841D01C565F071203E50317EA501F557DB6B9B71889F52BB53540274E3E48F7C005196,
            0x00EA6040E700403170DC5A51B1B140D5532777EE6651CECBE7223ECE0799C9DE5CF89984BFF76FE6B26BFEFA6EA16AFE,
        ),
        (
            0x00690D665D446F7BD960736BCBB2EFB4DE03ED7274B49A58E458C282F832D204F2CF88886D8C7C2EF094094409FD4DDF,
            0x00F8169FD28355189E549DA3151A70AA61EF11AC3D591BF12463B01ACEE304C24279B83F5E52270BD9A1CDD185EB8F93,
        ),
    ),
}


def f2mul(a, b, p, nr=1):
    return ((a[0] * b[0] - nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2inv(a, p, nr=1):
    ni = pow((a[0] * a[0] + nr * a[1] * a[1]) % p, -1, p)
    return (a[0] * ni % p, (-a[1]) * ni % p)


def g2_b(name, p):
    if name == "bn254":
        return f2mul((3, 0), f2inv((9, 1), p), p)
    if name == "bls12_377":  # reference curves/params/bls12_377.h weierstrass_b_g2_{re,im}
        return (0, 0x010222F6DB0FD6F343BD03737460C589DC7B4F91CD5FD889129207B63C6BF8000DD39E5C1CCCCCCD1C9ED9999999999A)
    return (4, 4)


def gen_curve_g2(name, fq, fr):
    p, _ = BIG[fq]
    nl = (p.bit_length() + 6 + RB - 1) // RB
    R = 1 << (RB * nl)
    _, gx, gy = G2[name]
    b = g2_b(name, p)
    nr = NONRES[fq]
    x3 = f2mul(f2mul(gx, gx, p, nr), gx, p, nr)
    y2 = f2mul(gy, gy, p, nr)
    assert ((y2[0] - x3[0] - b[0]) % p, (y2[1] - x3[1] - b[1]) % p) == (0, 0), name
    pair = lambda v: "{" + ", ".join(arr(limbs(c * R % p, nl)) for c in v) + "}"
    s = []
    s.append(f"struct {name}_g2 {{ // twist over Fq2 = Fq[u]/(u^2+{nr}), b' = {G2[name][0]}")
    s.append(f"  using fq = {fq}_params;")
    s.append(f"  using fr = {fr}_params;")
    s.append("  static constexpr int EXT_DEGREE = 2;")
    s.append(f"  static constexpr uint32_t B3[2][{nl}] = {pair((3 * b[0] % p, 3 * b[1] % p))}; // 3*b', Montgomery")
    s.append(f"  static constexpr uint32_t GX[2][{nl}] = {pair(gx)}; // generator, Montgomery")
    s.append(f"  static constexpr uint32_t GY[2][{nl}] = {pair(gy)};")
    s.append("};")
    return "\n".join(s)


def ec_add_aff(P, Q, p):
    """affine addition on y^2 = x^3 + b (None = identity); generator-time only"""
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        l = 3 * P[0] * P[0] * pow(2 * P[1], -1, p) % p
    else:
        l = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
    x = (l * l - P[0] - Q[0]) % p
    return x, (l * (P[0] - x) - P[1]) % p


def ec_mul_aff(k, P, p):
    R = None
    while k:
        if k & 1:
            R = ec_add_aff(R, P, p)
        P = ec_add_aff(P, P, p)
        k >>= 1
    return R


def glv_constants(p, r, gx, gy):
    """The GLV endomorphism of a j = 0 curve: phi(x, y) = (beta x, y) = lambda (x, y) on the r-torsion (beta^3 = 1 in Fq,
    lambda^2 + lambda + 1 = 0 mod r), a reduced basis (a1, b1), (a2, b2) of the lattice {(a, b): a + b lambda = 0 mod r} from the
    extended Euclidean algorithm on (r, lambda) (Guide to ECC, alg. 3.74), and the two rounding multipliers scaled by 2^256.
    k = k1 + k2 lambda with c_i = sign(G_i) ((k |G_i|) >> 256), k1 = k - c1 a1 - c2 a2, k2 = -c1 b1 - c2 b2, |k_i| < 2^129."""
    import math

    def roots(m):
        for g in range(2, 200):
            w = pow(g, (m - 1) // 3, m)
            if w != 1:
                return w, w * w % m
        raise AssertionError

    pair = None
    for beta in roots(p):
        for lam in roots(r):
            if ec_mul_aff(lam, (gx, gy), p) == (beta * gx % p, gy):
                pair = (beta, lam)
    assert pair
    beta, lam = pair
    rs = [(r, 1, 0), (lam, 0, 1)]
    while rs[-1][0] != 0:
        q = rs[-2][0] // rs[-1][0]
        rs.append((rs[-2][0] - q * rs[-1][0], rs[-2][1] - q * rs[-1][1], rs[-2][2] - q * rs[-1][2]))
    sq = math.isqrt(r)
    l = max(i for i, x in enumerate(rs) if x[0] >= sq)
    a1, b1 = rs[l + 1][0], -rs[l + 1][2]
    c1, c2 = (rs[l][0], -rs[l][2]), (rs[l + 2][0], -rs[l + 2][2])
    a2, b2 = c1 if c1[0] ** 2 + c1[1] ** 2 <= c2[0] ** 2 + c2[1] ** 2 else c2
    det = a1 * b2 - a2 * b1
    assert abs(det) == r and (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    sgn = 1 if det > 0 else -1
    g1 = (b2 * sgn << 256) // r if b2 * sgn >= 0 else -((-b2 * sgn << 256) // r)
    g2 = (-b1 * sgn << 256) // r if -b1 * sgn >= 0 else -((b1 * sgn << 256) // r)
    assert max(abs(x).bit_length() for x in (a1, b1, a2, b2)) <= 128 and max(abs(g1).bit_length(), abs(g2).bit_length()) <= 131
    return beta, lam, (a1, b1, a2, b2), (g1, g2)


def words(x, n):
    """n little-endian 32-bit words of x mod 2^(32 n) (two's complement for negative x)"""
    x %= 1 << (32 * n)
    return "{" + ", ".join(f"0x{(x >> (32 * i)) & 0xFFFFFFFF:08x}u" for i in range(n)) + "}"


def gen_curve(name, fq, fr, b, gx, gy):
    p, _ = BIG[fq]
    nl = (p.bit_length() + 6 + RB - 1) // RB
    R = 1 << (RB * nl)
    assert (gy * gy - gx * gx * gx - b) % p == 0
    s = []
    s.append(f"struct {name}_g1 {{")
    s.append(f"  using fq = {fq}_params;")
    s.append(f"  using fr = {fr}_params;")
    s.append("  static constexpr int EXT_DEGREE = 1;")
    s.append(f"  static constexpr uint32_t B3[{nl}] = {arr(limbs(3 * b % p * R % p, nl))}; // 3*b, Montgomery")
    s.append(f"  static constexpr uint32_t B3_SMALL = {3 * b if 0 < 3 * b < 64 else 0}; // 3*b as a plain integer when it is small (0: it is not), ec_dbl_quad.hpp")
    s.append(f"  static constexpr uint32_t GX[{nl}] = {arr(limbs(gx * R % p, nl))}; // generator, Montgomery")
    s.append(f"  static constexpr uint32_t GY[{nl}] = {arr(limbs(gy * R % p, nl))};")
    # GLV endomorphism (glv.hpp): every curve here has j = 0
    r, _ = BIG[fr]
    beta, lam, (a1, b1, a2, b2), (g1, g2) = glv_constants(p, r, gx, gy)
    s.append(f"  static constexpr uint32_t GLV_BETA[{nl}] = {arr(limbs(beta * R % p, nl))}; // phi(x, y) = (beta x, y) = lambda (x, y); Montgomery")
    s.append(f"  static constexpr uint32_t GLV_LAMBDA[8] = {words(lam, 8)}; // lambda^2 + lambda + 1 = 0 mod r (tests)")
    s.append(f"  static constexpr uint32_t GLV_A1[5] = {words(a1, 5)}, GLV_B1[5] = {words(b1, 5)}, GLV_A2[5] = {words(a2, 5)}, GLV_B2[5] = {words(b2, 5)}; // lattice basis, two's complement mod 2^160")
    s.append(f"  static constexpr uint32_t GLV_G1[5] = {words(abs(g1), 5)}, GLV_G2[5] = {words(abs(g2), 5)}; // |round-down(2^256 b2 / det)|, |... -b1 / det|")
    s.append(f"  static constexpr bool GLV_G1_NEG = {'true' if g1 < 0 else 'false'}, GLV_G2_NEG = {'true' if g2 < 0 else 'false'};")
    s.append("};")
    return "\n".join(s)


SMALL = {
    # name: (p, rou (order 2^two_adicity), ext nonresidue)
    "babybear": (0x78000001, 0x89),
    "koalabear": (0x7F000001, 0x6AC49F88),
}


def gen_small(name, p, rou):
    two_adicity = ((p - 1) & -(p - 1)).bit_length() - 1
    assert pow(rou, 1 << two_adicity, p) == 1 and pow(rou, 1 << (two_adicity - 1), p) != 1
    R = 1 << 32
    s = []
    s.append(f"struct {name}_params {{")
    s.append(f"  static constexpr uint32_t P = 0x{p:08x}u;")
    s.append(f"  static constexpr uint32_t PINV = 0x{pow(p, -1, R):08x}u;    // p^-1 mod 2^32")
    s.append(f"  static constexpr uint32_t NPINV = 0x{(-pow(p, -1, R)) % R:08x}u;   // -p^-1 mod 2^32")
    s.append(f"  static constexpr uint32_t ONE = 0x{R % p:08x}u;     // R mod p")
    s.append(f"  static constexpr uint32_t R2 = 0x{R * R % p:08x}u;      // R^2 mod p")
    s.append(f"  static constexpr uint32_t ROU = 0x{rou:08x}u;     // root of unity of order 2^TWO_ADICITY")
    s.append(f"  static constexpr int TWO_ADICITY = {two_adicity};")
    s.append("};")
    return "\n".join(s)


def main():
    out = []
    out.append("// GENERATED by tools/gen_consts.py -- do not edit.")
    out.append("// Constants for the 29-bit-radix Montgomery representation used on device (bigfield.hpp)")
    out.append("// and for the 31-bit NTT fields (smallfield.hpp).")
    out.append("#pragma once")
    out.append("#include <cstdint>")
    out.append("namespace icicle_hip {")
    out.append(f"static constexpr int RB = {RB};")
    out.append(f"static constexpr uint32_t RB_MASK = 0x{(1 << RB) - 1:08x}u;")
    for name, (p, l32) in BIG.items():
        out.append(gen_big(name, p, l32))
    for name, (fq, fr, b, gx, gy) in CURVES.items():
        out.append(gen_curve(name, fq, fr, b, gx, gy))
        if name in G2:
            out.append(gen_curve_g2(name, fq, fr))
    for name, (p, rou) in SMALL.items():
        out.append(gen_small(name, p, rou))
    out.append("} // namespace icicle_hip")
    text = "\n".join(out) + "\n"
    path = sys.argv[1] if len(sys.argv) > 1 else "icicle_amd/csrc/field_consts.h"
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path)


if __name__ == "__main__":
    main()
