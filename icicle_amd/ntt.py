"""ntt() / domain management: mirror of wrappers/rust/icicle-core/src/ntt/mod.rs:113-119,285-355."""
import ctypes
import numpy as np
from ._lib import lib, check, NTTConfigU32, NTTInitDomainConfig
from .runtime import DeviceVec

FORWARD, INVERSE = 0, 1
kNN, kNR, kRN, kRR, kNM, kMN = range(6)


def _ptr(x):
    if isinstance(x, DeviceVec):
        return x.ptr, True
    if isinstance(x, int):
        return x, True
    assert isinstance(x, np.ndarray) and x.dtype == np.uint32 and x.flags["C_CONTIGUOUS"]
    return x.ctypes.data, False


def get_root_of_unity(field: str, max_size: int) -> int:
    r = ctypes.c_uint32()
    check(getattr(lib, f"{field}_get_root_of_unity")(max_size, ctypes.byref(r)), "get_root_of_unity")
    return r.value


def get_root_of_unity_from_domain(field: str, logn: int) -> int:
    r = ctypes.c_uint32()
    check(getattr(lib, f"{field}_get_root_of_unity_from_domain")(logn, ctypes.byref(r)), "get_root_of_unity_from_domain")
    return r.value


def init_domain(field: str, primitive_root: int, cfg: NTTInitDomainConfig = None):
    cfg = cfg or NTTInitDomainConfig.default()
    r = ctypes.c_uint32(primitive_root)
    check(getattr(lib, f"{field}_ntt_init_domain")(ctypes.byref(r), ctypes.byref(cfg)), "ntt_init_domain")


def release_domain(field: str):
    check(getattr(lib, f"{field}_ntt_release_domain")(), "ntt_release_domain")


def ntt(field: str, inp, direction: int, cfg: NTTConfigU32 = None, out=None, size: int = None, extension: bool = False):
    cfg = cfg or NTTConfigU32.default()
    lanes = 4 if extension else 1
    ip, i_dev = _ptr(inp)
    cfg.are_inputs_on_device = i_dev
    if size is None:
        size = inp.size // (max(1, cfg.batch_size) * lanes)
    if out is None:
        out = np.zeros_like(inp)
    op, o_dev = _ptr(out)
    cfg.are_outputs_on_device = o_dev
    fn = getattr(lib, f"{field}_extension_ntt" if extension else f"{field}_ntt")
    check(fn(ip, size, direction, ctypes.byref(cfg), op), f"{field}_ntt")
    return out
