"""Runs in its OWN process (spawned by tests/test_gpu_rust_suite.py): the reference's Rust wrapper tests on the MSM / NTT /
ECNTT path, replayed CALL FOR CALL -- same allocation calls, same stream use, same order of copy and synchronize, same
config fields and ConfigExtension keys -- through the reference runtime + frontend libraries (oracle/_ref) with the HIP
plugin as the main device and the reference's "CPU" device as the ref device. tests/rustlike.py is the wrapper layer
(no Rust toolchain in this image); each function below cites the Rust function it replays.

usage: rust_suite_driver.py <check> <type> [reps]      prints 'RUST-REPLAY OK <check> <type>' on success
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import rustlike as R  # noqa: E402
from tests.rustlike import DeviceVec, HostSlice, IcicleStream, MSMConfig, NTTConfig, NTTInitDomainConfig  # noqa: E402

rng = np.random.default_rng(20260926)
MAX_SIZE = 1 << 18  # ntt/mod.rs:389, ecntt/mod.rs:121


def generate_random_affine_points_with_zeroes(P, size, num_zeroes):
    """msm/tests.rs:17-24 (A::zero() = (0, 0))"""
    points = P.generate_random_affine(size)
    for _ in range(num_zeroes):
        points[int(rng.integers(0, size))] = 0
    return points


def init_domain(F, max_size, fast_twiddles_mode):
    """ntt/tests.rs:17-27"""
    config = NTTInitDomainConfig()
    config.ext.set_bool(R.CUDA_NTT_FAST_TWIDDLES_MODE, fast_twiddles_mode)
    rou = F.get_root_of_unity(max_size)
    F.initialize_domain(rou, config)
    config.ext.drop()


def initialize_ntt(F):
    """impl_ntt_tests!::initialize (ntt/mod.rs:393-404) / impl_ecntt_tests!::initialize (ecntt/mod.rs:124-135)"""
    R.test_set_ref_device()
    init_domain(F, MAX_SIZE, False)
    R.test_set_main_device()
    init_domain(F, MAX_SIZE, False)
    R.test_set_main_device()


# ------------------------------------------------------------------------------------------------ MSM
def check_msm(P, reps=1):
    """msm/tests.rs:26-89. The order that matters: async msm() on a created stream into a DeviceVec, then the
    SYNCHRONOUS to_host_vec() (icicle_copy_to_host), and only then stream.synchronize()."""
    R.test_set_main_device()
    device_count = R.get_device_count()
    for device_id in range(device_count):  # (rayon par_iter over the devices; one here)
        test_sizes = [1, 5, 16, 32, 64, 128, 256, 1000, 1 << 18]
        R.test_set_main_device_with_id(device_id)
        stream = IcicleStream.create()
        msm_results = DeviceVec.device_malloc_async(1, P.PW, stream)
        for _ in range(reps):
            for test_size in test_sizes:
                points = generate_random_affine_points_with_zeroes(P, test_size, 2)
                scalars = P.scalar.generate_random(test_size)
                # (1) async msm on main device
                R.test_set_main_device_with_id(device_id)
                scalars_d = DeviceVec.device_malloc_async(test_size, P.SW, stream)
                scalars_d.copy_from_host_async(HostSlice(scalars), stream)
                P.scalar.to_mont(scalars_d, stream)  # convert to mont for testing MSM in this case
                cfg = MSMConfig()
                cfg.stream_handle = stream.handle
                cfg.are_scalars_montgomery_form = True
                cfg.is_async = True
                P.msm(scalars_d, HostSlice(points), cfg, msm_results)
                msm_host_result = msm_results.to_host_vec()  # <- BEFORE the synchronize
                stream.synchronize()
                # (2) compute on ref device and compare
                R.test_set_ref_device()
                ref_msm_host_result = P.zero(1)
                dflt = MSMConfig()
                P.msm(HostSlice(scalars), HostSlice(points), dflt, HostSlice(ref_msm_host_result))
                assert P.eq(ref_msm_host_result, msm_host_result), f"check_msm size {test_size}"
                cfg.ext.drop()
                dflt.ext.drop()
                scalars_d.drop()  # end of scope: Drop = icicle_free, with the ref device active (runtime.cpp:66-93)
        stream.destroy()
        R.test_set_main_device_with_id(device_id)
        msm_results.drop()


def _check_msm_batch(P, shared):
    """msm/tests.rs:91-172 (shared) / :174-254 (not shared)"""
    test_sizes = [1000, 1 << 14]
    batch_sizes = [1, 3, 1 << 4]
    stream = IcicleStream.create()
    precompute_factor = 8
    cfg = MSMConfig()
    cfg.stream_handle = stream.handle
    cfg.is_async = True
    cfg.ext.set_int(R.CUDA_MSM_LARGE_BUCKET_FACTOR, 5)
    cfg.c = 4
    R.warmup(stream)
    stream.synchronize()
    for test_size in test_sizes:
        R.test_set_main_device()
        if shared:
            cfg.precompute_factor = precompute_factor
            points = generate_random_affine_points_with_zeroes(P, test_size, 10)
            precomputed_points_d = DeviceVec.malloc(cfg.precompute_factor * test_size, P.AW)
            P.precompute_bases(HostSlice(points), cfg, precomputed_points_d)
        for batch_size in batch_sizes:
            nb = test_size if shared else test_size * batch_size
            if not shared:
                cfg.precompute_factor = precompute_factor
            scalars = P.scalar.generate_random(test_size * batch_size)
            scalars_h = HostSlice(scalars)
            if not shared:
                points = generate_random_affine_points_with_zeroes(P, test_size * batch_size, 10)
                precomputed_points_d = DeviceVec.malloc(cfg.precompute_factor * test_size * batch_size, P.AW)
                cfg.batch_size = batch_size
                cfg.are_points_shared_in_batch = False
                P.precompute_bases(HostSlice(points), cfg, precomputed_points_d)
            msm_results_1 = DeviceVec.malloc(batch_size, P.PW)
            msm_results_2 = DeviceVec.malloc(batch_size, P.PW)
            points_d = DeviceVec.malloc(nb, P.AW)
            points_d.copy_from_host_async(HostSlice(points), stream)
            cfg.precompute_factor = precompute_factor
            P.msm(scalars_h, precomputed_points_d, cfg, msm_results_1)
            cfg.precompute_factor = 1
            P.msm(scalars_h, points_d, cfg, msm_results_2)
            msm_host_result_1 = P.zero(batch_size)
            msm_host_result_2 = P.zero(batch_size)
            msm_results_1.copy_to_host_async(HostSlice(msm_host_result_1), stream)
            msm_results_2.copy_to_host_async(HostSlice(msm_host_result_2), stream)
            stream.synchronize()
            # (2) compute on ref device and compare to both cases (with or w/o precompute)
            R.test_set_ref_device()
            msm_ref_result = P.zero(batch_size)
            dflt = MSMConfig()
            P.msm(scalars_h, HostSlice(points), dflt, HostSlice(msm_ref_result))
            dflt.ext.drop()
            assert P.eq(msm_host_result_1, msm_ref_result), f"precomputed: size {test_size} batch {batch_size}"
            assert P.eq(msm_host_result_2, msm_ref_result), f"plain: size {test_size} batch {batch_size}"
            for v in (msm_results_1, msm_results_2, points_d):
                v.drop()  # (freed while the ref device is active, like the Rust scope end)
            if not shared:
                precomputed_points_d.drop()
            # The Rust loop leaves the REF device active here: its next iteration would allocate on "CPU" and hand the
            # main device's precomputed bases to msm() with the CPU device active, which the wrapper itself rejects
            # (msm/mod.rs:135-140) -- upstream this test only runs through where main == ref. The replay goes back to the
            # main device, which is what the test's own comment says it wants ("(1) compute MSM ... on main device").
            R.test_set_main_device()
        if shared:
            precomputed_points_d.drop()
    stream.destroy()
    cfg.ext.drop()


def check_msm_batch_shared(P, reps=1):
    _check_msm_batch(P, True)


def check_msm_batch_not_shared(P, reps=1):
    _check_msm_batch(P, False)


def check_msm_skewed_distributions(P, reps=1):
    """msm/tests.rs:256-304"""
    test_sizes = [1 << 10, 10000]
    test_threshold = 1 << 11
    batch_sizes = [1, 3, 1 << 4]
    for test_size in test_sizes:
        for batch_size in batch_sizes:
            tot = test_size * batch_size
            points = generate_random_affine_points_with_zeroes(P, tot, 100)
            scalars = P.scalar.zero(tot)
            for _ in range(tot):
                scalars[int(rng.integers(0, tot))] = P.scalar.one()
            for _ in range(test_threshold, test_size):
                scalars[int(rng.integers(0, tot))] = P.scalar.generate_random(1)[0]
            cfg = MSMConfig()
            if test_size < test_threshold:
                cfg.bitsize = 1
            R.test_set_main_device()
            msm_results = P.zero(batch_size)
            P.msm(HostSlice(scalars), HostSlice(points), cfg, HostSlice(msm_results))
            R.test_set_ref_device()
            msm_results_ref = P.zero(batch_size)
            P.msm(HostSlice(scalars), HostSlice(points), cfg, HostSlice(msm_results_ref))
            cfg.ext.drop()
            assert P.eq(msm_results, msm_results_ref), f"skewed: size {test_size} batch {batch_size}"


# ------------------------------------------------------------------------------------------------ NTT
def check_ntt(F, reps=1):
    """ntt/tests.rs:37-87"""
    for test_size in (1 << 4, 1 << 17):
        scalars = F.generate_random(test_size)
        ntt_result_main = F.zero(test_size)
        ntt_result_ref = F.zero(test_size)
        config = NTTConfig(F)
        for alg in (R.NTT_RADIX2, R.NTT_MIXED_RADIX):
            config.ext.set_int(R.CUDA_NTT_ALGORITHM, alg)
            R.test_set_main_device()
            F.ntt(HostSlice(scalars), R.kForward, config, HostSlice(ntt_result_main))
            R.test_set_ref_device()
            F.ntt(HostSlice(scalars), R.kForward, config, HostSlice(ntt_result_ref))
            assert np.array_equal(ntt_result_main, ntt_result_ref)
            R.test_set_main_device()
            F.ntt_inplace(HostSlice(ntt_result_main), R.kInverse, config)
            R.test_set_ref_device()
            F.ntt_inplace(HostSlice(ntt_result_ref), R.kInverse, config)
            assert np.array_equal(ntt_result_main, ntt_result_ref)
        config.ext.drop()


def check_ntt_coset_from_subgroup(F, reps=1):
    """ntt/tests.rs:89-157"""
    for test_size in (1 << 4, 1 << 16):
        R.test_set_main_device()
        small_size = test_size >> 1
        test_size_rou = F.get_root_of_unity(test_size)
        scalars = F.generate_random(small_size)
        for alg in (R.NTT_RADIX2, R.NTT_MIXED_RADIX):
            config = NTTConfig(F)
            config.ordering = R.kNR
            config.ext.set_int(R.CUDA_NTT_ALGORITHM, alg)
            ntt_result_half = F.zero(small_size)
            ntt_result_coset = F.zero(small_size)
            scalars_h = HostSlice(np.ascontiguousarray(scalars[:small_size]))
            F.ntt(scalars_h, R.kForward, config, HostSlice(ntt_result_half))
            assert not np.array_equal(ntt_result_half, scalars[:small_size])
            config.coset_gen = test_size_rou
            F.ntt(scalars_h, R.kForward, config, HostSlice(ntt_result_coset))
            R.test_set_ref_device()
            ntt_coset_ref = F.zero(small_size)
            F.ntt(scalars_h, R.kForward, config, HostSlice(ntt_coset_ref))
            assert np.array_equal(ntt_result_coset, ntt_coset_ref)
            R.test_set_main_device()
            ntt_large_result = F.zero(test_size)
            config.coset_gen = F.one()
            if scalars.shape[0] < test_size:  # scalars.resize(test_size, F::zero())
                scalars = np.ascontiguousarray(np.concatenate([scalars, F.zero(test_size - scalars.shape[0])]))
            F.ntt(HostSlice(scalars), R.kForward, config, HostSlice(ntt_large_result))
            assert np.array_equal(ntt_result_half, ntt_large_result[:small_size])
            assert np.array_equal(ntt_result_coset, ntt_large_result[small_size:])
            config.coset_gen = test_size_rou
            config.ordering = R.kRN
            intt_result = F.zero(small_size)
            F.ntt(HostSlice(ntt_result_coset), R.kInverse, config, HostSlice(intt_result))
            assert np.array_equal(intt_result, scalars[:small_size])
            config.ext.drop()


def check_ntt_coset_interpolation_nm(F, reps=1):
    """ntt/tests.rs:159-213"""
    for test_size in (1 << 9, 1 << 10, 1 << 11, 1 << 13, 1 << 14, 1 << 16):
        test_size_rou = F.get_root_of_unity(test_size << 1)
        coset_generators = [test_size_rou, F.generate_random(1)[0]]
        scalars = F.generate_random(test_size)
        for coset_gen in coset_generators:
            config = NTTConfig(F)
            config.ordering = R.kNM
            config.ext.set_int(R.CUDA_NTT_ALGORITHM, R.NTT_MIXED_RADIX)
            R.test_set_main_device()
            intt_result = F.zero(test_size)
            F.ntt(HostSlice(scalars), R.kInverse, config, HostSlice(intt_result))
            R.test_set_ref_device()
            intt_result_ref = F.zero(test_size)
            F.ntt(HostSlice(scalars), R.kInverse, config, HostSlice(intt_result_ref))
            config.coset_gen = coset_gen
            config.ordering = R.kMN
            R.test_set_main_device()
            coset_evals = F.zero(test_size)
            F.ntt(HostSlice(intt_result), R.kForward, config, HostSlice(coset_evals))
            R.test_set_ref_device()
            coset_evals_ref = F.zero(test_size)
            F.ntt(HostSlice(intt_result_ref), R.kForward, config, HostSlice(coset_evals_ref))
            assert np.array_equal(coset_evals, coset_evals_ref), f"nm: size {test_size}"
            config.ext.drop()


def check_ntt_arbitrary_coset(F, reps=1):
    """ntt/tests.rs:215-253"""
    for test_size in (1 << 4, 1 << 17):
        coset_generators = [F.generate_random(1)[0], F.get_root_of_unity(test_size), F.one()]
        for coset_gen in coset_generators:
            scalars = F.generate_random(test_size)
            scalars_ref = scalars.copy()
            config = NTTConfig(F)
            config.coset_gen = coset_gen
            for alg in (R.NTT_RADIX2, R.NTT_MIXED_RADIX):
                config.ordering = R.kNR
                config.ext.set_int(R.CUDA_NTT_ALGORITHM, alg)
                R.test_set_main_device()
                F.ntt_inplace(HostSlice(scalars), R.kForward, config)
                R.test_set_ref_device()
                F.ntt_inplace(HostSlice(scalars_ref), R.kForward, config)
                assert np.array_equal(scalars, scalars_ref)
                config.ordering = R.kRN
                R.test_set_main_device()
                F.ntt_inplace(HostSlice(scalars), R.kInverse, config)
                R.test_set_ref_device()
                F.ntt_inplace(HostSlice(scalars_ref), R.kInverse, config)
                assert np.array_equal(scalars, scalars_ref)
            config.ext.drop()


def check_ntt_batch(F, reps=1):
    """ntt/tests.rs:255-340: everything on the MAIN device -- a batch == its single transforms, and
    transpose -> columns_batch transform -> transpose == the row batch"""
    R.test_set_main_device()
    for test_size in (1 << 4, 1 << 12):
        coset_generators = [F.one(), F.generate_random(1)[0]]
        config = NTTConfig(F)
        for batch_size in (1, 1 << 4, 100):
            scalars_a = F.generate_random(test_size * batch_size)
            scalars = HostSlice(scalars_a)
            for coset_gen in coset_generators:
                for is_inverse in (R.kInverse, R.kForward):
                    for ordering in (R.kNN, R.kNR, R.kRN, R.kRR, R.kNM, R.kMN):
                        config.coset_gen = coset_gen
                        config.ordering = ordering
                        batch_ntt_result = F.zero(batch_size * test_size)
                        for alg in (R.NTT_RADIX2, R.NTT_MIXED_RADIX):
                            config.batch_size = batch_size
                            config.ext.set_int(R.CUDA_NTT_ALGORITHM, alg)
                            F.ntt(scalars, is_inverse, config, HostSlice(batch_ntt_result))
                            config.batch_size = 1
                            one_ntt_result = F.ones(test_size)
                            for i in range(batch_size):
                                F.ntt(scalars[i * test_size:(i + 1) * test_size], is_inverse, config, HostSlice(one_ntt_result))
                                assert np.array_equal(batch_ntt_result[i * test_size:(i + 1) * test_size], one_ntt_result)
                        nof_rows, nof_cols = batch_size, test_size
                        config.batch_size = batch_size
                        config.columns_batch = True
                        transposed_input = F.zero(batch_size * test_size)
                        F.matrix_transpose(scalars, nof_rows, nof_cols, HostSlice(transposed_input))
                        col_batch_ntt_result = F.zero(batch_size * test_size)
                        F.ntt(HostSlice(transposed_input), is_inverse, config, HostSlice(col_batch_ntt_result))
                        F.matrix_transpose(HostSlice(col_batch_ntt_result), nof_cols, nof_rows, HostSlice(transposed_input))
                        assert np.array_equal(batch_ntt_result, transposed_input), (test_size, batch_size, is_inverse, ordering)
                        config.columns_batch = False
        config.ext.drop()


def check_ntt_device_async(F, reps=1):
    """ntt/tests.rs:342-420: fast-twiddles init_domain on an already initialised device, CUDA_NTT_ALGORITHM ext, in-place
    async transform of a DeviceVec on a created stream + async copy to the host"""
    R.test_set_main_device()
    device_count = R.get_device_count()
    for device_id in range(device_count):
        R.test_set_main_device_with_id(device_id)
        stream = IcicleStream.create()
        init_domain(F, 1 << 16, True)
        config = NTTConfig(F)
        for test_size in (1 << 4, 1 << 12):
            coset_generators = [F.one(), F.generate_random(1)[0]]
            for batch_size in (1, 1 << 4, 100):
                scalars = F.generate_random(test_size * batch_size)
                scalars_d = DeviceVec.from_host_slice(scalars)
                for coset_gen in coset_generators:
                    for ordering in (R.kNN, R.kRR):
                        config.coset_gen = coset_gen
                        config.ordering = ordering
                        config.batch_size = batch_size
                        config.is_async = False
                        config.stream_handle = None
                        scalars_clone = scalars.copy()
                        R.test_set_ref_device()
                        F.ntt_inplace(HostSlice(scalars_clone), R.kForward, config)
                        R.test_set_main_device_with_id(device_id)
                        config.is_async = True
                        config.stream_handle = stream.handle
                        for alg in (R.NTT_RADIX2, R.NTT_MIXED_RADIX):
                            config.ext.set_int(R.CUDA_NTT_ALGORITHM, alg)
                            ntt_result_h = F.zero(test_size * batch_size)
                            F.ntt_inplace(scalars_d, R.kForward, config)
                            scalars_d.copy_to_host_async(HostSlice(ntt_result_h), stream)
                            stream.synchronize()
                            assert np.array_equal(scalars_clone, ntt_result_h), (test_size, batch_size, ordering)
                            F.ntt_inplace(scalars_d, R.kInverse, config)
                            scalars_d.copy_to_host_async(HostSlice(ntt_result_h), stream)
                            stream.synchronize()
                            assert np.array_equal(scalars, ntt_result_h)
                scalars_d.drop()
        config.ext.drop()
        stream.destroy()


def check_ntt_async_copy_before_sync(F, reps=20):
    """NOT a Rust test: the order of check_msm (sync copy BEFORE stream.synchronize) applied to an async NTT, as VERDICT r04
    item 1 asks: async in-place transform of a DeviceVec on a created stream, to_host_vec(), then synchronize."""
    R.test_set_main_device()
    stream = IcicleStream.create()
    config = NTTConfig(F)
    config.is_async = True
    config.stream_handle = stream.handle
    for _ in range(reps):
        for test_size, batch_size in ((1 << 4, 1), (1 << 12, 100), (1 << 17, 2)):
            scalars = F.generate_random(test_size * batch_size)
            R.test_set_ref_device()
            exp = scalars.copy()
            rc = NTTConfig(F)
            rc.batch_size = batch_size
            F.ntt_inplace(HostSlice(exp), R.kForward, rc)
            rc.ext.drop()
            R.test_set_main_device()
            config.batch_size = batch_size
            scalars_d = DeviceVec.device_malloc_async(test_size * batch_size, F.W, stream)
            scalars_d.copy_from_host_async(HostSlice(scalars), stream)
            F.ntt_inplace(scalars_d, R.kForward, config)
            got = scalars_d.to_host_vec()
            stream.synchronize()
            assert np.array_equal(got, exp), (test_size, batch_size)
            scalars_d.drop()
    config.ext.drop()
    stream.destroy()


def check_release_domain(F, reps=1):
    """ntt/tests.rs:422-433"""
    R.test_set_main_device()
    F.release_domain()
    R.test_set_ref_device()
    F.release_domain()


# ------------------------------------------------------------------------------------------------ ECNTT
def check_ecntt(P, reps=1):
    """ecntt/tests.rs:11-44"""
    for test_size in (1 << 4, 1 << 9, 1 << 11):
        for d in (R.kForward, R.kInverse):
            config = NTTConfig(P.scalar)
            points = P.generate_random_projective(test_size)
            ecntt_result = P.zero(test_size)
            ecntt_result_ref = P.zero(test_size)
            R.test_set_main_device()
            P.ecntt(HostSlice(points), d, config, HostSlice(ecntt_result))
            R.test_set_ref_device()
            P.ecntt(HostSlice(points), d, config, HostSlice(ecntt_result_ref))
            assert P.eq(ecntt_result, ecntt_result_ref), (test_size, d)
            inv_dir = R.kInverse if d == R.kForward else R.kForward
            R.test_set_main_device()
            P.ecntt_inplace(HostSlice(ecntt_result), inv_dir, config)
            assert P.eq(ecntt_result, points)
            config.ext.drop()


def check_ecntt_batch(P, reps=1):
    """ecntt/tests.rs:46-90"""
    R.test_set_main_device()
    for test_size in (1 << 4, 1 << 9):
        config = NTTConfig(P.scalar)
        for batch_size in (1, 1 << 4, 21):
            points_a = P.generate_random_projective(test_size * batch_size)
            points = HostSlice(points_a)
            for is_inverse in (R.kInverse, R.kForward):
                config.ordering = R.kNN
                batch_ntt_result = P.zero(batch_size * test_size)
                config.batch_size = batch_size
                P.ecntt(points, is_inverse, config, HostSlice(batch_ntt_result))
                config.batch_size = 1
                one_ntt_result = P.zero(test_size)
                for i in range(batch_size):
                    P.ecntt(points[i * test_size:(i + 1) * test_size], is_inverse, config, HostSlice(one_ntt_result))
                    assert P.eq(batch_ntt_result[i * test_size:(i + 1) * test_size], one_ntt_result), (test_size, batch_size, i)
        config.ext.drop()


MSM_CHECKS = {f.__name__: f for f in (check_msm, check_msm_batch_shared, check_msm_batch_not_shared, check_msm_skewed_distributions)}
NTT_CHECKS = {f.__name__: f for f in (check_ntt, check_ntt_coset_from_subgroup, check_ntt_coset_interpolation_nm, check_ntt_arbitrary_coset,
                                      check_ntt_batch, check_ntt_device_async, check_ntt_async_copy_before_sync, check_release_domain)}
ECNTT_CHECKS = {f.__name__: f for f in (check_ecntt, check_ecntt_batch)}


def main():
    check, tname = sys.argv[1], sys.argv[2]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    # The frontend libraries first, the backend after -- the order of a real binary, which LINKS libicicle_curve_<c> /
    # libicicle_field_<f> and loads backends at run time. (ctypes opens libraries RTLD_LOCAL: were the field library first
    # brought in by a plugin DSO, its CpuNttDomain<S>::s_ntt_domain -- a weak template static also emitted into the curve
    # library for cpu_ecntt -- would bind to another copy than the curve library's, and the reference's CPU ecntt would see an
    # uninitialised domain. Test-harness artefact, nothing of the product.)
    is_curve = tname in R.ref.CURVE_LIMBS
    T = R.Curve(tname) if is_curve else R.ScalarField(tname, 1)
    if os.environ.get("RUST_REPLAY_MAIN") == "CPU":  # self-test of the replay code without a GPU: main == ref == "CPU"
        R.MAIN_DEVICE = ("CPU", 0)
    else:
        R.test_load_and_init_devices(os.path.join(ROOT, "plugin", "lib", "backend", "hip"))
    R.test_set_main_device()
    kw = {} if reps is None else {"reps": reps}
    if check in MSM_CHECKS:
        MSM_CHECKS[check](T, **kw)
    elif check in ECNTT_CHECKS:
        initialize_ntt(T.scalar)
        ECNTT_CHECKS[check](T, **kw)
    else:
        F = T.scalar if is_curve else T
        initialize_ntt(F)
        NTT_CHECKS[check](F, **kw)
    print(f"RUST-REPLAY OK {check} {tname}")


if __name__ == "__main__":
    main()
