"""GPU parity: <curve>_ecntt through the C ABI vs the reference CPU backend (which runs its generic NTT with
E = projective_t) and the O(N^2) definition. Projective representatives depend on the order of additions, so
the comparison is on to_affine() limbs, as for the MSM (icicle/tests/test_curve_api.cpp ecntt test compares
main vs reference device results the same way)."""
import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import cached_points, from_words, points_to_array, rand_scalars, to_words

pytestmark = pytest.mark.gpu
CURVES = ["bn254", "bls12_381", "bls12_377"]
DOMAIN_LOG = 12


def to_projective(C, pts):
    """affine python points -> projective_t words; identity -> (0 : 1 : 0)"""
    L = C.limbs_q
    rows = []
    for p in pts:
        if p == pyref.INF:
            rows.append(np.concatenate([to_words([0], L)[0], to_words([1], L)[0], to_words([0], L)[0]]))
        else:
            rows.append(np.concatenate([to_words([p[0]], L)[0], to_words([p[1]], L)[0], to_words([1], L)[0]]))
    return np.ascontiguousarray(np.stack(rows).astype(np.uint32))


def affine_list(refc, C, proj_words, n):
    L = C.limbs_q
    aff = refc.to_affine(proj_words.reshape(n, 3 * L))
    return [(from_words(a[:L]), from_words(a[L:])) for a in aff]


@pytest.fixture(scope="module", params=CURVES)
def env(request, hip):
    from icicle_amd import ntt as N

    cname = request.param
    F = pyref.NTT_FIELDS[cname]
    sf = ref.RefScalarNttField(cname)
    root = N.get_root_of_unity(cname, 1 << DOMAIN_LOG)
    N.init_domain(cname, root)
    sf.init_domain(root)
    yield cname, pyref.CURVES[cname], F, ref.RefCurve(cname), N
    N.release_domain(cname)
    sf.release_domain()


def test_ecntt_vs_definition(env, hip):
    cname, C, F, refc, N = env
    for logn in (0, 1, 3):
        n = 1 << logn
        pts = list(cached_points(C, n))
        if n >= 4:
            pts[2] = pyref.INF
        x = to_projective(C, pts).reshape(-1)
        y = N.ecntt(cname, x, N.FORWARD)
        assert affine_list(refc, C, y, n) == pyref.ecntt_naive(C, F, pts, pyref.omega(F, logn))
        back = N.ecntt(cname, y, N.INVERSE)
        assert affine_list(refc, C, back, n) == pts
        cfg = hip.NTTConfigU256.default()
        cfg.ordering = N.kRR
        cfg.set_coset_gen(7)
        z = N.ecntt(cname, x, N.INVERSE, cfg)
        assert affine_list(refc, C, z, n) == pyref.ecntt_naive(C, F, pts, pyref.omega(F, logn), inverse=True, coset_gen=7, ordering="RR")


@pytest.mark.parametrize("logn", [2, 5, 8, 10])
def test_ecntt_matrix_vs_reference(env, hip, logn):
    cname, C, F, refc, N = env
    rng = np.random.default_rng(500 + logn)
    n = 1 << logn
    L = C.limbs_q
    for trial in range(3 if logn < 10 else 1):
        batch = int(rng.choice([1, 2, 3])) if logn < 10 else 1
        columns = bool(rng.integers(0, 2))
        ordering = int(rng.integers(0, 6))
        direction = int(rng.integers(0, 2))
        coset = 1 if rng.integers(0, 2) else rand_scalars(rng, 1, F.p)[0] or 3
        base = refc.generate_affine_points(n * batch)  # period-100 repetition: equal points and P + P occur
        proj = np.concatenate([base, np.tile(to_words([1], L), (n * batch, 1))], axis=1).astype(np.uint32)
        x = np.ascontiguousarray(proj).reshape(-1)
        cfg = hip.NTTConfigU256.default()
        cfg.batch_size, cfg.columns_batch, cfg.ordering = batch, columns, ordering
        cfg.set_coset_gen(coset)
        got = N.ecntt(cname, x, direction, cfg)
        exp = refc.ecntt(x, n, direction, batch=batch, columns_batch=columns, ordering=ordering, coset_gen=coset)
        assert np.array_equal(refc.to_affine(got.reshape(-1, 3 * L)), refc.to_affine(exp.reshape(-1, 3 * L))), \
            (cname, logn, batch, columns, ordering, direction, hex(coset))


def test_ecntt_device_and_errors(env, hip):
    cname, C, F, refc, N = env
    from icicle_amd.runtime import DeviceVec

    n = 64
    L = C.limbs_q
    pts = list(cached_points(C, n))
    x = to_projective(C, pts).reshape(-1)
    d_in, d_out = DeviceVec.from_host(x), DeviceVec.from_host(np.zeros_like(x))
    N.ecntt(cname, d_in, N.FORWARD, out=d_out, size=n)
    y = d_out.to_host()
    assert np.array_equal(refc.to_affine(y.reshape(n, 3 * L)), refc.to_affine(refc.ecntt(x, n, 0).reshape(n, 3 * L)))
    with pytest.raises(hip.IcicleError):
        N.ecntt(cname, x, N.FORWARD, size=48)  # not a power of two
    with pytest.raises(hip.IcicleError):
        N.ecntt(cname, x, N.FORWARD, size=1 << (DOMAIN_LOG + 1))  # larger than the domain


@pytest.mark.parametrize("radix_log", [1, 2, 3, 4, 5])
def test_ecntt_every_stage_radix_vs_reference(hip, radix_log):
    """the radix-2^r matrix-form stages (icicle_amd/csrc/ecntt.hip: k_ecntt_terms / k_ecntt_sums) with r forced through
    ICICLE_HIP_ECNTT_RADIX_LOG (read once per process, hence a child process): 2^7 and 2^9 points -- uneven stage widths, e.g.
    r = 4: 4 + 3 and 3 + 3 + 3 -- forward and inverse, a batch, bit-reversed orderings and a coset, against the reference CPU backend"""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import icicle_amd as hip
from icicle_amd import ntt as N, runtime
from oracle import pyref, ref
from tests.util import to_words
runtime.set_device(0)
cname = "bn254"
C, F = pyref.CURVES[cname], pyref.NTT_FIELDS[cname]
refc, sf = ref.RefCurve(cname), ref.RefScalarNttField(cname)
root = N.get_root_of_unity(cname, 1 << 9)
N.init_domain(cname, root); sf.init_domain(root)
L = C.limbs_q
for logn, batch, columns, ordering, direction, coset in ((7, 1, False, 0, 0, 1), (7, 3, True, 3, 1, 5), (9, 1, False, 1, 1, 1), (9, 2, False, 2, 0, 11)):
    n = 1 << logn
    base = refc.generate_affine_points(n * batch)
    x = np.ascontiguousarray(np.concatenate([base, np.tile(to_words([1], L), (n * batch, 1))], axis=1).astype(np.uint32)).reshape(-1)
    cfg = hip.NTTConfigU256.default()
    cfg.batch_size, cfg.columns_batch, cfg.ordering = batch, columns, ordering
    cfg.set_coset_gen(coset)
    got = N.ecntt(cname, x, direction, cfg)
    exp = refc.ecntt(x, n, direction, batch=batch, columns_batch=columns, ordering=ordering, coset_gen=coset)
    assert np.array_equal(refc.to_affine(got.reshape(-1, 3 * L)), refc.to_affine(exp.reshape(-1, 3 * L))), (logn, batch, columns, ordering, direction, coset)
print("RADIX OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["ICICLE_HIP_ECNTT_RADIX_LOG"] = str(radix_log)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RADIX OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_ecntt_timing_vs_reference_cpu(hip):
    """ECNTT 2^10 .. 2^16 timed on the GPU and on the reference CPU backend (the box's host
    cores), results equal as group elements; the numbers are appended to gpurun_out/ecntt_timing.txt (copied into profiles/)."""
    import os
    import time

    from icicle_amd import ntt as N

    cname = "bn254"
    C = pyref.CURVES[cname]
    L = C.limbs_q
    refc = ref.RefCurve(cname)
    sf = ref.RefScalarNttField(cname)
    root = N.get_root_of_unity(cname, 1 << 16)
    N.init_domain(cname, root)
    sf.init_domain(root)
    lines = []
    try:
        for logn in (10, 12, 14, 16):
            n = 1 << logn
            base = refc.generate_affine_points(n)
            x = np.ascontiguousarray(np.concatenate([base, np.tile(to_words([1], L), (n, 1))], axis=1).astype(np.uint32)).reshape(-1)
            N.ecntt(cname, x, N.FORWARD)  # warm
            t0 = time.perf_counter()
            got = N.ecntt(cname, x, N.FORWARD)
            t_gpu = time.perf_counter() - t0
            t0 = time.perf_counter()
            exp = refc.ecntt(x, n, 0)
            t_cpu = time.perf_counter() - t0
            assert np.array_equal(refc.to_affine(got.reshape(-1, 3 * L)), refc.to_affine(exp.reshape(-1, 3 * L)))
            lines.append(f"ecntt {cname} 2^{logn}: GPU {t_gpu * 1e3:9.2f} ms (host in/out)   reference CPU backend {t_cpu * 1e3:9.2f} ms on {os.cpu_count()} threads   x{t_cpu / t_gpu:.1f}")
    finally:
        N.release_domain(cname)
        sf.release_domain()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "ecntt_timing.txt"), "a") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


@pytest.mark.parametrize("curve_id,cname", [(0, "bn254"), (1, "bls12_381"), (2, "bls12_377")])
def test_quad_cooperative_group_operations_against_the_one_lane_formulas(hip, curve_id, cname):
    """the butterflies' addition (four product rounds over a DPP quad, ec_dbl_quad.hpp EcQuadAdd; the five-round ec.hpp add_quad) and
    doubling (two product levels, EcDblSmallB::dbl_quad) on every pair (a G, +- b G), a, b = 0..6 -- P = Q, P = -Q, O + P, P + O, O + O and
    runs of five doublings included -- against the complete one-lane formulas (the reference's, projective.h:73-143; checked against
    Python integers in tests/test_host_math.py), compared as group elements on the device; every lane of a quad must hold the result"""
    import ctypes
    from icicle_amd._lib import lib, check

    bad = ctypes.c_int(-1)
    check(lib.icicle_hip_selftest_quad_group_ops(curve_id, ctypes.byref(bad)), "selftest")
    assert bad.value == 0, (cname, bad.value)


def test_ecntt_2_14_mixed_stage_plan_vs_reference(hip):
    """2^14 points: the first size whose stage plan mixes radix-4 matrix-form stages with radix-2 ones under the 32768-quad budget
    (ecntt.hip ecntt_run) -- forward, and inverse with a bit-reversed ordering and a coset, against the reference CPU backend"""
    from icicle_amd import ntt as N

    cname, logn = "bn254", 14
    C, F = pyref.CURVES[cname], pyref.NTT_FIELDS[cname]
    refc, sf = ref.RefCurve(cname), ref.RefScalarNttField(cname)
    n, L = 1 << logn, C.limbs_q
    root = N.get_root_of_unity(cname, n)
    N.init_domain(cname, root)
    sf.init_domain(root)
    try:
        base = refc.generate_affine_points(n)
        x = np.ascontiguousarray(np.concatenate([base, np.tile(to_words([1], L), (n, 1))], axis=1).astype(np.uint32)).reshape(-1)
        for direction, ordering, coset in ((0, 0, 1), (1, 2, 7)):
            cfg = hip.NTTConfigU256.default()
            cfg.ordering = ordering
            cfg.set_coset_gen(coset)
            got = N.ecntt(cname, x, direction, cfg)
            exp = refc.ecntt(x, n, direction, ordering=ordering, coset_gen=coset)
            assert np.array_equal(refc.to_affine(got.reshape(-1, 3 * L)), refc.to_affine(exp.reshape(-1, 3 * L))), (direction, ordering, coset)
    finally:
        N.release_domain(cname)
        sf.release_domain()
