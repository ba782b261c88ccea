"""GPU: the reference's Rust wrapper tests on the MSM / NTT / ECNTT path, one pytest case per Rust test function, each
replayed call for call (tests/rust_suite_driver.py over tests/rustlike.py) through the reference's unmodified runtime +
frontend libraries with the HIP plugin as the main device and the reference "CPU" device as the ref device.

  wrappers/rust/icicle-core/src/msm/tests.rs    check_msm :26-89 (x50: copy BEFORE synchronize), check_msm_batch_shared :91-172,
                                                check_msm_batch_not_shared :174-254, check_msm_skewed_distributions :256-304
  wrappers/rust/icicle-core/src/ntt/tests.rs    check_ntt :37-87, _coset_from_subgroup :89-157, _coset_interpolation_nm :159-213,
                                                _arbitrary_coset :215-253, _batch :255-340, _device_async :342-420,
                                                check_release_domain :422-433
  wrappers/rust/icicle-core/src/ecntt/tests.rs  check_ecntt :11-44, check_ecntt_batch :46-90

Each case runs in its own process so that the reference runtime owns the process-wide icicle_* symbols, as in a user's binary.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUG = os.path.join(ROOT, "plugin", "lib", "backend", "hip", "libicicle_backend_hip_device.so")

CASES = [
    ("check_msm", "bn254", 50), ("check_msm", "bls12_381", 5),
    ("check_msm_batch_shared", "bn254", None), ("check_msm_batch_shared", "bls12_381", None),
    ("check_msm_batch_not_shared", "bn254", None), ("check_msm_batch_not_shared", "bls12_381", None),
    ("check_msm_skewed_distributions", "bn254", None), ("check_msm_skewed_distributions", "bls12_381", None),
    ("check_ntt", "babybear", None), ("check_ntt", "koalabear", None), ("check_ntt", "bn254", None),
    ("check_ntt_coset_from_subgroup", "babybear", None), ("check_ntt_coset_from_subgroup", "bn254", None),
    ("check_ntt_coset_interpolation_nm", "babybear", None), ("check_ntt_coset_interpolation_nm", "koalabear", None),
    ("check_ntt_coset_interpolation_nm", "bls12_381", None),
    ("check_ntt_arbitrary_coset", "babybear", None), ("check_ntt_arbitrary_coset", "koalabear", None), ("check_ntt_arbitrary_coset", "bn254", None),
    ("check_ntt_batch", "babybear", None), ("check_ntt_batch", "bn254", None),
    ("check_ntt_device_async", "babybear", None), ("check_ntt_device_async", "koalabear", None), ("check_ntt_device_async", "bn254", None),
    ("check_ntt_async_copy_before_sync", "babybear", 20), ("check_ntt_async_copy_before_sync", "bn254", 5),
    ("check_release_domain", "babybear", None), ("check_release_domain", "bn254", None),
    ("check_ecntt", "bn254", None), ("check_ecntt", "bls12_381", None),
    ("check_ecntt_batch", "bn254", None),
]


@pytest.mark.parametrize("check,tname,reps", CASES, ids=[f"{c}-{t}" for c, t, _ in CASES])
def test_rust_wrapper_test_replayed_on_the_hip_backend(hip, check, tname, reps):
    if not os.path.exists(PLUG):
        pytest.skip("plugin not built (plugin/build_plugin.sh needs /root/reference)")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rust_suite_driver.py"), check, tname] + ([str(reps)] if reps else [])
    env = dict(os.environ)
    env.pop("RUST_REPLAY_MAIN", None)
    env.pop("ICICLE_HIP_STREAMS_NONBLOCKING", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and f"RUST-REPLAY OK {check} {tname}" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
