"""icicle_amd: MI355X (gfx950) backend for ICICLE's MSM / NTT hot path.

The product is libicicle_hip.so (hand-written HIP, C ABI in include/icicle_hip.h); this package is
the thin host-side mirror of the reference's wrapper API (wrappers/rust/icicle-core/src/{msm,ntt},
icicle-runtime) used by the tests and bench.py. There is no CPU fallback anywhere in here.
"""
from ._lib import IcicleError, Device, MSMConfig, NTTConfigU32, NTTConfigU64, NTTConfigU256, NTTInitDomainConfig, VecOpsConfig, lib, LIB_PATH  # noqa: F401
from . import runtime, msm, ntt, vecops  # noqa: F401

__all__ = ["runtime", "msm", "ntt", "vecops", "VecOpsConfig", "IcicleError", "Device", "MSMConfig", "NTTConfigU32", "NTTConfigU64", "NTTConfigU256", "NTTInitDomainConfig"]
