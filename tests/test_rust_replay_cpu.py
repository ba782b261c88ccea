"""CPU: the replay code itself (tests/rust_suite_driver.py, tests/rustlike.py) run with main == ref == the reference's "CPU"
device, so that a Python slip in the replay shows here and not on the GPU box. check_msm_batch_not_shared is left out: the
reference's CPU backend extends only nof_bases = len / batch_size points (cpu_msm.hpp:470-485) while the wrapper hands it
batch_size times as many, so upstream's own test cannot pass with main == "CPU" for batch_size > 1."""
import os
import subprocess
import sys

import pytest

from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("check_msm", "bn254"), ("check_msm_batch_shared", "bn254"), ("check_msm_skewed_distributions", "bn254"),
         ("check_ntt", "babybear"), ("check_ntt_coset_from_subgroup", "koalabear"), ("check_ntt_coset_interpolation_nm", "babybear"),
         ("check_ntt_arbitrary_coset", "bn254"), ("check_ntt_device_async", "babybear"), ("check_ntt_async_copy_before_sync", "babybear"),
         ("check_release_domain", "babybear"), ("check_ecntt", "bn254")]


@pytest.mark.parametrize("check,tname", CASES, ids=[f"{c}-{t}" for c, t in CASES])
def test_replay_driver_self_test_on_the_cpu_device(check, tname):
    if not ref.available("device") or not ref.available(tname):
        pytest.skip("oracle/_ref not built")
    env = dict(os.environ, RUST_REPLAY_MAIN="CPU")
    args = [check, tname] + (["2"] if check == "check_ntt_async_copy_before_sync" else [])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rust_suite_driver.py")] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and f"RUST-REPLAY OK {check} {tname}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
