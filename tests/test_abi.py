"""CPU: the C-ABI library loads and exports every symbol include/icicle_hip.h declares, and the
config structs have the reference's layout (probe-verified sizes)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    """every function the header declares, after macro expansion by the C preprocessor"""
    import subprocess

    text = subprocess.check_output(["gcc", "-E", "-P", os.path.join(ROOT, "include", "icicle_hip.h")], text=True)
    names = set(re.findall(r"\b(?:icicle_error_t|const char\s*\*|void|int|_Bool|icicle_config_extension_t\s*\*)\s+(\w+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from icicle_amd import _lib

    syms = declared_symbols()
    assert len(syms) >= 55, syms
    for s in syms:
        assert hasattr(_lib.lib, s), f"libicicle_hip.so does not export {s}"
    # and the binding table in _lib.py covers the header
    assert set(syms) == set(_lib.RUNTIME_SYMBOLS + _lib.API_SYMBOLS)


def test_struct_layouts_match_reference():
    from icicle_amd import _lib

    assert ctypes.sizeof(_lib.Device) == 68 and _lib.Device.id.offset == 64
    M = _lib.MSMConfig
    assert ctypes.sizeof(M) == 40
    assert [getattr(M, f).offset for f, _ in M._fields_] == [0, 8, 12, 16, 20, 24, 25, 26, 27, 28, 29, 30, 32]
    N = _lib.NTTConfigU32
    assert ctypes.sizeof(N) == 40
    assert [getattr(N, f).offset for f, _ in N._fields_] == [0, 8, 12, 16, 20, 24, 25, 26, 32]
    assert ctypes.sizeof(_lib.NTTInitDomainConfig) == 24
    assert ctypes.sizeof(_lib.DeviceProperties) == 12


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device every compute entry point must fail (no silent CPU path)."""
    import numpy as np
    import icicle_amd
    from icicle_amd import msm, ntt, runtime

    if runtime.get_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(icicle_amd.IcicleError):
        runtime.set_device(0)
    with pytest.raises(icicle_amd.IcicleError):
        msm.msm("bn254", np.zeros((4, 8), np.uint32), np.zeros((4, 16), np.uint32))
    with pytest.raises(icicle_amd.IcicleError):
        ntt.init_domain("babybear", 0x89)


def test_host_side_helpers_without_gpu():
    from icicle_amd import ntt
    from oracle import pyref

    for fname, F in (("babybear", pyref.BABYBEAR), ("koalabear", pyref.KOALABEAR)):
        for logn in (0, 1, 5, F.two_adicity):
            assert ntt.get_root_of_unity(fname, 1 << logn) == pyref.omega(F, logn)
        import icicle_amd
        with pytest.raises(icicle_amd.IcicleError):
            ntt.get_root_of_unity(fname, 1 << (F.two_adicity + 1))


def test_config_extension_roundtrip():
    from icicle_amd._lib import lib

    e = lib.create_config_extension()
    lib.config_extension_set_int(e, b"hip_num_devices", 21)
    lib.config_extension_set_bool(e, b"fast_twiddles", True)  # foreign (CUDA) key: tolerated
    assert lib.config_extension_get_int(e, b"hip_num_devices") == 21
    assert lib.config_extension_get_bool(e, b"fast_twiddles") is True
    c = lib.clone_config_extension(e)
    assert lib.config_extension_get_int(c, b"hip_num_devices") == 21
    lib.destroy_config_extension(e)
    lib.destroy_config_extension(c)
