"""Runs in its OWN process (spawned by tests/test_gpu_plugin.py): drives the HIP backend the way a user
of the reference does -- reference runtime + reference frontend libraries, backend found by
icicle_load_backend(), device selected by icicle_set_device({"HIP",0}) -- and checks it against the
reference's "CPU" device in the same process (the reference's own differential pattern,
icicle/tests/test_base.h:22-63, test_curve_api.cpp:36-79, test_mod_arithmetic_api.h:614-695,
test_device_api.cpp:17-259). Prints 'PLUGIN OK' on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref, ref  # noqa: E402
from tests.util import cached_points, points_to_array, rand_scalars, to_words  # noqa: E402


def main():
    rt = ref.RefRuntime()
    curve = ref.RefCurve("bn254")
    bls = ref.RefCurve("bls12_381")
    field = ref.RefNttField("babybear")
    koala = ref.RefNttField("koalabear")
    assert "HIP" not in rt.registered_devices()
    assert rt.load_backend(os.path.join(ROOT, "plugin", "lib", "backend", "hip")) == 0
    devs = rt.registered_devices()
    assert "HIP" in devs and "CPU" in devs, devs

    # ---- device API behaviour (test_device_api.cpp) ----
    assert rt.set_device("HIP", 10) == 1          # INVALID_DEVICE, thread keeps its device (:183-189)
    assert rt.set_device("HIP", 0) == 0
    assert rt.get_device_count() >= 1
    rc, _ = rt.malloc((1 << 64) - 1)
    assert rc != 0                                # :157-165
    rc, d = rt.malloc(1 << 20)
    assert rc == 0
    assert rt.lib.icicle_is_active_device_memory(d) == 0 and rt.lib.icicle_is_host_memory(d) != 0
    a = np.arange(4096, dtype=np.uint32)
    b = np.zeros_like(a)
    assert rt.to_device(d + 1024, a) == 0         # interior pointers (:76-96)
    rc2, d2 = rt.malloc(1 << 20)
    assert rt.copy(d2, d + 1024, a.nbytes) == 0   # D2D, direction inferred from the tracker (:53-74)
    assert rt.copy(b.ctypes.data, d2, a.nbytes) == 0
    assert np.array_equal(a, b)
    assert rt.lib.icicle_memset(d2, 0, a.nbytes) == 0
    rt.to_host(b, d2)
    assert not b.any()
    assert rt.free(d) == 0 and rt.free(d2) == 0

    # ---- MSM: HIP vs CPU through the reference frontend ----
    rng = np.random.default_rng(42)
    for cobj, C in ((curve, pyref.BN254), (bls, pyref.BLS12_381)):
        n = 3000
        pts = list(cached_points(C, n))
        pts[7] = pyref.INF
        bases = points_to_array(C, pts)
        sc = to_words(rand_scalars(rng, n * 2, C.r), 8)
        assert rt.set_device("HIP", 0) == 0
        got = cobj.msm(sc, bases, batch=2)
        pre = cobj.precompute_bases(bases, 4)
        got_pre = cobj.msm(sc, pre, batch=2, precompute_factor=4)
        assert rt.set_device("CPU", 0) == 0
        exp = cobj.msm(sc, bases, batch=2)
        assert np.array_equal(cobj.to_affine(got), cobj.to_affine(exp))
        assert np.array_equal(cobj.to_affine(got_pre), cobj.to_affine(exp))
        for bidx in range(2):
            assert cobj.projective_eq(got[bidx], exp[bidx]) and cobj.is_on_curve(got[bidx])

    # ---- the Rust MSM test flow through the reference frontend: scalars converted to Montgomery form ON
    # the main device (wrappers/rust/icicle-core/src/msm/tests.rs:54-59), then msm with the flag set
    C = pyref.BN254
    n = 2000
    bases = points_to_array(C, cached_points(C, n))
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    assert rt.set_device("HIP", 0) == 0
    scm = ref.ref_convert_montgomery("bn254", "bn254_scalar_convert_montgomery", sc, n, True)
    got = curve.msm(scm, bases, scalars_mont=True)
    am = ref.ref_convert_montgomery("bn254", "bn254_affine_convert_montgomery", bases, n, True)
    assert rt.set_device("CPU", 0) == 0
    assert np.array_equal(scm, ref.ref_convert_montgomery("bn254", "bn254_scalar_convert_montgomery", sc, n, True))
    assert np.array_equal(am, ref.ref_convert_montgomery("bn254", "bn254_affine_convert_montgomery", bases, n, True))
    assert np.array_equal(curve.to_affine(got), curve.to_affine(curve.msm(sc, bases)))

    # ---- NTT: HIP (device buffers from the reference's icicle_malloc) vs CPU ----
    for fobj, F in ((field, pyref.BABYBEAR), (koala, pyref.KOALABEAR)):
        logn, batch = 14, 3
        n = 1 << logn
        root = fobj.get_root_of_unity(1 << (logn + 2))
        x = rng.integers(0, F.p, size=n * batch, dtype=np.uint32)
        assert rt.set_device("CPU", 0) == 0
        fobj.init_domain(root)
        exp_f = fobj.ntt(x, n, 0, batch=batch)
        exp_i = fobj.ntt(x, n, 1, batch=batch, ordering=2, coset_gen=5)
        exp_e = fobj.ntt(x[: 4 * 256], 256, 0, extension=True)
        assert rt.set_device("HIP", 0) == 0
        fobj.init_domain(root)
        assert fobj.get_root_of_unity_from_domain(logn) == pyref.omega(F, logn)
        assert np.array_equal(fobj.ntt(x, n, 0, batch=batch), exp_f)
        assert np.array_equal(fobj.ntt(x, n, 1, batch=batch, ordering=2, coset_gen=5), exp_i)
        assert np.array_equal(fobj.ntt(x[: 4 * 256], 256, 0, extension=True), exp_e)
        # element-wise vec-ops registered next to the NTT: dispatched to "HIP" by the reference frontend
        import ctypes

        def vec2(op, a, b):
            cfg = ref.VecOpsConfig(None, False, False, False, False, 1, False, None)
            out = np.zeros_like(b)
            fn = getattr(fobj.lib, f"{fobj.name}_{op}")
            if op == "bit_reverse":
                fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
                assert fn(b.ctypes.data, b.size, ctypes.byref(cfg), out.ctypes.data) == 0
            else:
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
                assert fn(a.ctypes.data, b.ctypes.data, b.size, ctypes.byref(cfg), out.ctypes.data) == 0
            return out

        va, vb = x[:n].copy(), x[n:2 * n].copy()
        hip_res = [vec2(op, va, vb) for op in ("vector_add", "vector_sub", "vector_mul", "bit_reverse")]
        assert rt.set_device("CPU", 0) == 0
        cpu_res = [vec2(op, va, vb) for op in ("vector_add", "vector_sub", "vector_mul", "bit_reverse")]
        assert all(np.array_equal(h, c) for h, c in zip(hip_res, cpu_res))
        assert rt.set_device("HIP", 0) == 0
        # the Rust suite's check_ntt_batch (wrappers/rust/icicle-core/src/ntt/tests.rs:255-340), everything on the MAIN
        # device: a batch == its single transforms, and transpose -> columns_batch transform -> transpose == the batch
        assert rt.set_device("HIP", 0) == 0
        for test_size in (1 << 4, 1 << 12):
            for bsz in (1, 1 << 4, 100):
                sc_in = rng.integers(0, F.p, size=test_size * bsz, dtype=np.uint32)
                for coset in (1, int(rng.integers(2, F.p))):
                    for direction in (1, 0):
                        for ordering in range(6):
                            batch_res = fobj.ntt(sc_in, test_size, direction, batch=bsz, ordering=ordering, coset_gen=coset)
                            for i in (0, bsz // 2, bsz - 1):
                                one = fobj.ntt(sc_in[i * test_size:(i + 1) * test_size].copy(), test_size, direction, ordering=ordering, coset_gen=coset)
                                assert np.array_equal(batch_res[i * test_size:(i + 1) * test_size], one)
                            t_in = fobj.matrix_transpose(sc_in, bsz, test_size)
                            col_res = fobj.ntt(t_in, test_size, direction, batch=bsz, columns_batch=True, ordering=ordering, coset_gen=coset)
                            back = fobj.matrix_transpose(col_res, test_size, bsz)
                            assert np.array_equal(batch_res, back), (fobj.name, test_size, bsz, coset, direction, ordering)
        # matrix_transpose itself: batched, in place, extension elements; "HIP" vs "CPU"
        m = rng.integers(0, F.p, size=3 * 37 * 50 * 4, dtype=np.uint32)
        hip_t = [fobj.matrix_transpose(m, 37, 200, batch=3), fobj.matrix_transpose(m.copy(), 64, 128, batch=1, inplace=True)[: 64 * 128],
                 fobj.matrix_transpose(m, 37, 50, batch=3, extension=True)]
        assert rt.set_device("CPU", 0) == 0
        cpu_t = [fobj.matrix_transpose(m, 37, 200, batch=3), fobj.matrix_transpose(m.copy(), 64, 128, batch=1, inplace=True)[: 64 * 128],
                 fobj.matrix_transpose(m, 37, 50, batch=3, extension=True)]
        assert all(np.array_equal(h, c) for h, c in zip(hip_t, cpu_t))
        assert rt.set_device("HIP", 0) == 0
        rc, d_in = rt.malloc(x.nbytes)
        rc, d_out = rt.malloc(x.nbytes)
        rt.to_device(d_in, x)
        assert fobj.ntt_device(d_in, d_out, n, 0, batch=batch) == 0
        y = np.zeros_like(x)
        rt.to_host(y, d_out)
        assert np.array_equal(y, exp_f)
        assert fobj.ntt_device(d_out, d_out, n, 1, batch=batch) == 0  # in place on device
        rt.to_host(y, d_out)
        assert np.array_equal(y, x)
        rt.free(d_in)
        rt.free(d_out)
        fobj.release_domain()
        assert rt.set_device("CPU", 0) == 0
        fobj.release_domain()

    # ---- G2 MSM (+ G2 Montgomery conversion) on "HIP" vs "CPU" through the reference frontend ----
    for name in ("bn254", "bls12_381"):
        C2 = pyref.G2_CURVES[name]
        g2 = ref.RefCurve(name, g2=True)
        n = 1500
        bases = g2.generate_affine_points(n)
        sc = to_words(rand_scalars(rng, n * 2, C2.base.r), 8)
        assert rt.set_device("HIP", 0) == 0
        got = g2.msm(sc, bases, batch=2)
        pre = g2.precompute_bases(bases, 3, c=8)
        am = ref.ref_convert_montgomery(name, f"{name}_g2_affine_convert_montgomery", bases, n, True)
        assert rt.set_device("CPU", 0) == 0
        exp = g2.msm(sc, bases, batch=2)
        assert np.array_equal(g2.to_affine(got), g2.to_affine(exp))
        assert np.array_equal(pre, g2.precompute_bases(bases, 3, c=8))
        assert np.array_equal(am, ref.ref_convert_montgomery(name, f"{name}_g2_affine_convert_montgomery", bases, n, True))
        for bidx in range(2):
            assert g2.projective_eq(got[bidx], exp[bidx]) and g2.is_on_curve(got[bidx])

    # ---- NTT over the curves' 256-bit scalar fields: "HIP" vs "CPU" through the reference frontend ----
    for name in ("bn254", "bls12_381"):
        F = pyref.NTT_FIELDS[name]
        sf = ref.RefScalarNttField(name)
        logn, batch = 12, 2
        n = 1 << logn
        root = sf.get_root_of_unity(1 << (logn + 1))
        x = to_words(rand_scalars(rng, n * batch, F.p), 8).reshape(-1)
        assert rt.set_device("CPU", 0) == 0
        sf.init_domain(root)
        exp_f = sf.ntt(x, n, 0, batch=batch)
        exp_i = sf.ntt(x, n, 1, batch=batch, ordering=3, coset_gen=7, columns_batch=True)
        assert rt.set_device("HIP", 0) == 0
        sf.init_domain(root)
        assert sf.get_root_of_unity_from_domain(logn) == pyref.omega(F, logn)
        assert np.array_equal(sf.ntt(x, n, 0, batch=batch), exp_f)
        assert np.array_equal(sf.ntt(x, n, 1, batch=batch, ordering=3, coset_gen=7, columns_batch=True), exp_i)
        # ECNTT on the same domain: "HIP" vs "CPU"
        Cc = pyref.CURVES[name]
        cobj = ref.RefCurve(name)
        m = 64
        proj = np.concatenate([cobj.generate_affine_points(m), np.tile(to_words([1], Cc.limbs_q), (m, 1))], axis=1)
        proj = np.ascontiguousarray(proj.astype(np.uint32)).reshape(-1)
        got_e = cobj.ecntt(proj, m, 0, ordering=1, coset_gen=3)
        assert rt.set_device("CPU", 0) == 0
        exp_e = cobj.ecntt(proj, m, 0, ordering=1, coset_gen=3)
        assert np.array_equal(cobj.to_affine(got_e.reshape(m, -1)), cobj.to_affine(exp_e.reshape(m, -1)))
        assert rt.set_device("HIP", 0) == 0
        sf.release_domain()
        assert rt.set_device("CPU", 0) == 0
        sf.release_domain()
    print("PLUGIN OK")


if __name__ == "__main__":
    main()
