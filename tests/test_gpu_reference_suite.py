"""GPU: the reference's OWN test sources, compiled unmodified (oracle/build_ref_tests.sh, GoogleTest stand-in
oracle/shim/gtest) and run with the HIP plugin as the main device -- the reference's test base makes the first
registered non-CPU device the device under test and "CPU" the reference (icicle/tests/test_base.h:37-46), and every
test compares the two (ASSERT_EQ on group elements for MSM / ECNTT, memcmp for NTT and vector ops).

  test_device_api.cpp            all of DeviceApiTest (SetDefaultDevice, MemoryCopyAsync with malloc_async / free_async,
                                 memoryTracker with 200 live allocations, ...)
  test_curve_api.cpp             CurveApiTest.{msm, msm_pre_compute, msmCpuThreads, msm_bitsize, msmG2, MontConversion*,
                                 ecntt, ecnttDeviceMem} + the host-arithmetic CurveSanity suite, for bn254, bls12_381 and
                                 bls12_377; the MSM part of it for grumpkin
  test_mod_arithmetic_api.h      ModArithTest.{ntt, montgomeryConversion} for the base and the extension field,
                                 {vectorVectorOps, bitReverse} for the base field, ModArithTestBase.scalarVectorOps,
                                 for babybear, koalabear, goldilocks, stark252 and the three pairing curves' scalar fields
Each binary runs in its own process (the reference runtime owns the process-wide icicle_* symbols) and several times:
the reference tests draw their sizes / orderings / cosets from a time-seeded generator."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, "oracle", "_ref", "tests")
BACKEND = os.path.join(ROOT, "plugin", "lib", "backend")


def _run(binary, gfilter, repeat=1, timeout=900):
    exe = os.path.join(TESTS, binary)
    if not os.path.exists(exe) or not os.path.isdir(os.path.join(BACKEND, "hip")):
        pytest.skip("reference tests / plugin not built (oracle/build_ref_tests.sh and plugin/build_plugin.sh need /root/reference)")
    env = dict(os.environ)
    env["ICICLE_BACKEND_INSTALL_DIR"] = BACKEND
    outs = []
    for _ in range(repeat):
        r = subprocess.run([exe, f"--gtest_filter={gfilter}"], capture_output=True, text=True, timeout=timeout, env=env)
        out = r.stdout + r.stderr
        outs.append(out)
        assert "Main-device=HIP" in out, out[-3000:]  # the device under test really is this backend
        assert r.returncode == 0 and "[  FAILED  ]" not in out, out[-6000:]
    return outs


def _ran(out, name):
    return f"[       OK ] {name}" in out


def test_reference_device_api_suite(hip):
    out = _run("test_device_api", "*")[0]
    for t in ("UnregisteredDeviceError", "SetDefaultDevice", "MemoryCopySync", "MemoryCopySyncWithOffset", "MemoryCopyAsync",
              "CopyDeviceInference", "Memset", "ApiError", "InvalidDevice", "memoryTracker"):
        assert _ran(out, f"DeviceApiTest.{t}"), (t, out[-3000:])


@pytest.mark.parametrize("curve", ["bn254", "bls12_381", "bls12_377", "grumpkin"])
def test_reference_curve_api_suite(hip, curve):
    outs = _run(f"test_curve_api_{curve}", "CurveApiTest.*:CurveSanity*", repeat=2, timeout=1500)
    names = ["msm", "msm_pre_compute", "msmCpuThreads", "MontConversionAffine", "MontConversionProjective", "msm_bitsize"]
    if curve != "grumpkin":  # (the reference builds Grumpkin with an MSM only: no G2, no ECNTT)
        names += ["msmG2", "MontConversionG2Affine", "MontConversionG2Projective", "ecntt", "ecnttDeviceMem"]
    for t in names:
        assert _ran(outs[0], f"CurveApiTest.{t}"), (t, outs[0][-3000:])


@pytest.mark.parametrize("field", ["babybear", "koalabear", "goldilocks", "bn254", "bls12_381", "bls12_377", "stark252"])
def test_reference_modarith_suite(hip, field):
    ext = field in ("babybear", "koalabear", "goldilocks")
    flt = "ModArithTest/*.ntt:ModArithTest/*.montgomeryConversion:ModArithTest/0.vectorVectorOps:ModArithTest/0.bitReverse:ModArithTestBase.scalarVectorOps"
    outs = _run(f"test_modarith_{field}", flt, repeat=6)  # 6 random draws of (logn, batch, layout, ordering, coset, in-place)
    names = ["ModArithTest/0.ntt", "ModArithTest/0.montgomeryConversion", "ModArithTest/0.vectorVectorOps", "ModArithTest/0.bitReverse",
             "ModArithTestBase.scalarVectorOps"] + (["ModArithTest/1.ntt", "ModArithTest/1.montgomeryConversion"] if ext else [])
    for t in names:
        assert _ran(outs[0], t), (t, outs[0][-3000:])


@pytest.mark.parametrize("field", ["babybear", "koalabear", "goldilocks", "bn254"])
def test_reference_matrix_transpose_suite(hip, field):
    """MatrixTest.matrixTranspose (icicle/tests/test_matrix_api.h:450-505): 3 x (128 x 256) out of place and in place on every
    registered device, memcmp'd with a host loop -- for scalar_t and, where the field has one, extension_t"""
    out = _run(f"test_matrix_{field}", "MatrixTest/*.matrixTranspose")[0]
    names = ["MatrixTest/0.matrixTranspose"] + (["MatrixTest/1.matrixTranspose"] if field != "bn254" else [])
    for t in names:
        assert _ran(out, t), (t, out[-3000:])


@pytest.mark.parametrize("example", ["msm", "ntt", "best_practice_ntt"])
def test_reference_cpp_examples_on_hip(hip, example):
    """examples/c++/msm, examples/c++/ntt and examples/c++/best-practice-ntt (uploads, downloads and in-place NTTs on three
    streams at once) of the reference, compiled unmodified, with device "HIP" selected by name
    (examples_utils.h try_load_and_set_backend_device): the drop-in as a user of the reference sees it. The examples
    print their timings and throw (non-zero exit) on any API error."""
    exe = os.path.join(TESTS, f"example_{example}")
    if not os.path.exists(exe):
        pytest.skip("examples not built")
    env = dict(os.environ)
    env["ICICLE_BACKEND_INSTALL_DIR"] = BACKEND
    r = subprocess.run([exe, "HIP"], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "selecting HIP device" in out and "falling back to CPU" not in out, out[-3000:]
