"""CPU: the background reference pool of the full-size GPU tests (tests/refpool.py) returns exactly what the inline reference
calls return -- MSM jobs on in-process lanes, NTT jobs (with a chained second call) through the worker process."""
import numpy as np
import pytest

from oracle import pyref, ref
from tests.refpool import RefPool
from tests.util import cached_points, points_to_array, rand_scalars, to_words


@pytest.mark.skipif(not (ref.available("bn254") and ref.available("babybear")), reason="oracle/_ref not built")
def test_pool_results_equal_inline_reference_calls():
    rng = np.random.default_rng(11)
    C = pyref.BN254
    n = 300
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    F = pyref.BABYBEAR
    x = rng.integers(0, F.p, size=4 << 10, dtype=np.uint32)
    pool = RefPool()
    try:
        pool.submit_msm("m1", "bn254", sc, bases)
        pool.submit_msm("m2", "bn254", sc[:100].copy(), bases[:100].copy(), lane="other")
        pool.submit_ntt("n1", "babybear", x, 10, 0, batch=4, lane="a")
        pool.submit_ntt("n2", "babybear", x[:1 << 10], 10, 0, ordering=1, coset_gen=31, chain=[(1, 2, 31)], lane="a")  # kNR forward, kRN inverse
        pool.submit_ntt("n3", "babybear", x[:1 << 9], 9, 1, lane="b")
        pool.start()
        refc = ref.RefCurve("bn254")
        assert np.array_equal(pool.result("m1"), refc.msm(sc, bases))
        assert np.array_equal(pool.result("m2"), refc.msm(sc[:100].copy(), bases[:100].copy()))
        rf = ref.RefNttField("babybear")
        rf.init_domain(rf.get_root_of_unity(1 << 10))
        try:
            assert np.array_equal(pool.result("n1", timeout=120), rf.ntt(x, 1 << 10, 0, batch=4))
            e, back = pool.result("n2", timeout=120)
            assert np.array_equal(e, rf.ntt(x[:1 << 10], 1 << 10, 0, ordering=1, coset_gen=31))
            assert np.array_equal(back, x[:1 << 10])
        finally:
            rf.release_domain()
        rf.init_domain(rf.get_root_of_unity(1 << 9))
        try:
            assert np.array_equal(pool.result("n3", timeout=120), rf.ntt(x[:1 << 9].copy(), 1 << 9, 1))
        finally:
            rf.release_domain()
        assert set(pool.timings) >= {"m1", "m2", "n1", "n2", "n3"}
    finally:
        pool.close()


def test_full_size_tests_join_the_pool_last():
    """the collection hook moves every refjob test behind the rest of the session (conftest.py) and every key a test names has a starter"""
    import importlib

    for mod in ("tests.test_gpu_msm", "tests.test_gpu_msm_distributions", "tests.test_gpu_ntt_fullsize", "tests.test_gpu_fullsize_configs"):
        m = importlib.import_module(mod)
        assert m.REF_JOBS, mod
        for k, (prio, fn) in m.REF_JOBS.items():
            assert callable(fn) and isinstance(prio, int), (mod, k)
        for name in dir(m):
            t = getattr(m, name)
            for mk in getattr(t, "pytestmark", []) if name.startswith("test_") else []:
                if mk.name == "refjob":
                    assert set(mk.args) <= set(m.REF_JOBS), (mod, name, mk.args)
