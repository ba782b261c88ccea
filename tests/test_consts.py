"""CPU: the generated device constants (icicle_amd/csrc/field_consts.h) are derived from the published
moduli; when /root/reference is present, cross-check every modulus / generator / root of unity against
the reference headers (the GPU box has no /root/reference: skipped there)."""
import os
import re

import pytest

from oracle import pyref

REF = "/root/reference/icicle/include/icicle"


def _limbs(text, name):
    m = re.search(name + r"\s*=\s*\{([^}]*)\}", text)
    vals = [int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    return sum(v << (32 * i) for i, v in enumerate(vals))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not available")
def test_constants_match_reference_headers():
    t = open(f"{REF}/fields/snark_fields/bn254_base.h").read()
    assert _limbs(t, r"storage<8> modulus") == pyref.BN254.q
    t = open(f"{REF}/fields/snark_fields/bn254_scalar.h").read()
    assert _limbs(t, r"storage<8> modulus") == pyref.BN254.r
    assert _limbs(t, r"storage<8> rou") == pyref.BN254_FR.rou
    t = open(f"{REF}/fields/snark_fields/bls12_381_base.h").read()
    assert _limbs(t, r"storage<12> modulus") == pyref.BLS12_381.q
    t = open(f"{REF}/fields/snark_fields/bls12_381_scalar.h").read()
    assert _limbs(t, r"storage<8> modulus") == pyref.BLS12_381.r
    t = open(f"{REF}/curves/params/bls12_381.h").read()
    assert _limbs(t, r"point_field_t gen_x") == pyref.BLS12_381.gx
    assert _limbs(t, r"point_field_t gen_y") == pyref.BLS12_381.gy
    t = open(f"{REF}/curves/params/bn254.h").read()
    assert _limbs(t, r"point_field_t gen_x") == 1 and _limbs(t, r"point_field_t gen_y") == 2
    assert _limbs(t, r"point_field_t weierstrass_b") == 3
    t = open(f"{REF}/fields/snark_fields/bls12_377_base.h").read()
    assert _limbs(t, r"storage<12> modulus") == pyref.BLS12_377.q
    assert re.search(r"nonresidue = 5;", t) and re.search(r"nonresidue_is_negative = true", t)  # Fq2 = Fq[u]/(u^2 + 5)
    t = open(f"{REF}/fields/snark_fields/bls12_377_scalar.h").read()
    assert _limbs(t, r"storage<8> modulus") == pyref.BLS12_377.r
    assert _limbs(t, r"storage<8> rou") == pyref.BLS12_377_FR.rou
    t = open(f"{REF}/curves/params/bls12_377.h").read()
    assert _limbs(t, r"point_field_t gen_x") == pyref.BLS12_377.gx and _limbs(t, r"point_field_t gen_y") == pyref.BLS12_377.gy
    assert _limbs(t, r"point_field_t weierstrass_b") == pyref.BLS12_377.b == 1
    g2 = pyref.BLS12_377_G2
    assert (_limbs(t, r"point_field_t g2_gen_x_re"), _limbs(t, r"point_field_t g2_gen_x_im")) == g2.gx
    assert (_limbs(t, r"point_field_t g2_gen_y_re"), _limbs(t, r"point_field_t g2_gen_y_im")) == g2.gy
    assert (_limbs(t, r"point_field_t weierstrass_b_g2_re"), _limbs(t, r"point_field_t weierstrass_b_g2_im")) == g2.b
    t = open(f"{REF}/curves/params/grumpkin.h").read()  # base field = bn254's scalar field and vice versa (grumpkin_{base,scalar}.h)
    assert _limbs(t, r"point_field_t gen_x") == pyref.GRUMPKIN.gx and _limbs(t, r"point_field_t gen_y") == pyref.GRUMPKIN.gy
    assert re.search(r"is_b_neg = true", t) and _limbs(t, r"\n\s*static constexpr point_field_t weierstrass_b") == pyref.GRUMPKIN.q - pyref.GRUMPKIN.b == 17
    assert "bn254::fp_config fq_config" in open(f"{REF}/fields/snark_fields/grumpkin_base.h").read()
    assert "bn254::fq_config fp_config" in open(f"{REF}/fields/snark_fields/grumpkin_scalar.h").read()
    t = open(f"{REF}/fields/stark_fields/stark252.h").read()
    assert _limbs(t, r"storage<8> modulus") == pyref.STARK252.p and _limbs(t, r"storage<8> rou") == pyref.STARK252.rou
    for f, h in ((pyref.BABYBEAR, "babybear"), (pyref.KOALABEAR, "koalabear")):
        t = open(f"{REF}/fields/stark_fields/{h}.h").read()
        assert _limbs(t, r"storage<1> modulus") == f.p
        assert _limbs(t, r"storage<1> rou") == f.rou


def test_generated_header_is_current(tmp_path):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "field_consts.h"
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_consts.py"), str(out)])
    assert out.read_text() == open(os.path.join(root, "icicle_amd", "csrc", "field_consts.h")).read()


def test_curve_and_field_parameters_are_consistent():
    for c in (pyref.BN254, pyref.BLS12_381, pyref.BLS12_377, pyref.GRUMPKIN):
        g = (c.gx, c.gy)
        assert pyref.on_curve(c, g)
        assert pyref.ec_mul(c, c.r - 1, g) == pyref.ec_neg(c, g)  # r*G = identity
    g2 = pyref.BLS12_377_G2
    assert pyref.g2_on_curve(g2, (g2.gx, g2.gy)) and pyref.g2_mul(g2, g2.base.r - 1, (g2.gx, g2.gy)) == pyref.g2_neg(g2, (g2.gx, g2.gy))
    for f in (pyref.BABYBEAR, pyref.KOALABEAR, pyref.BN254_FR, pyref.BLS12_377_FR, pyref.STARK252):
        assert pow(f.rou, 1 << f.two_adicity, f.p) == 1 and pow(f.rou, 1 << (f.two_adicity - 1), f.p) == f.p - 1
