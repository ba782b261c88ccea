"""ntt() / domain management: mirror of wrappers/rust/icicle-core/src/ntt/mod.rs:113-119,285-355."""
import ctypes
import numpy as np
from ._lib import lib, check, NTTConfigU32, NTTConfigU64, NTTConfigU256, NTTInitDomainConfig, SCALAR_NTT_FIELDS, GOLD
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


def _is_big(field: str) -> bool:
    """multi-word elements: roots travel as word arrays (goldilocks reads / writes the first two of the eight)"""
    return field in SCALAR_NTT_FIELDS or field == GOLD


def _words(x: int):
    return (ctypes.c_uint32 * 8)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def _int(w) -> int:
    return sum(int(w[i]) << (32 * i) for i in range(8))


def get_root_of_unity(field: str, max_size: int) -> int:
    if _is_big(field):
        w = (ctypes.c_uint32 * 8)()
        check(getattr(lib, f"{field}_get_root_of_unity")(max_size, w), "get_root_of_unity")
        return _int(w)
    r = ctypes.c_uint32()
    check(getattr(lib, f"{field}_get_root_of_unity")(max_size, ctypes.byref(r)), "get_root_of_unity")
    return r.value


def get_root_of_unity_from_domain(field: str, logn: int) -> int:
    if _is_big(field):
        w = (ctypes.c_uint32 * 8)()
        check(getattr(lib, f"{field}_get_root_of_unity_from_domain")(logn, w), "get_root_of_unity_from_domain")
        return _int(w)
    r = ctypes.c_uint32()
    check(getattr(lib, f"{field}_get_root_of_unity_from_domain")(logn, ctypes.byref(r)), "get_root_of_unity_from_domain")
    return r.value


def init_domain(field: str, primitive_root: int, cfg: NTTInitDomainConfig = None):
    cfg = cfg or NTTInitDomainConfig.default()
    if _is_big(field):
        check(getattr(lib, f"{field}_ntt_init_domain")(_words(primitive_root), ctypes.byref(cfg)), "ntt_init_domain")
        return
    r = ctypes.c_uint32(primitive_root)
    check(getattr(lib, f"{field}_ntt_init_domain")(ctypes.byref(r), ctypes.byref(cfg)), "ntt_init_domain")


def release_domain(field: str):
    check(getattr(lib, f"{field}_ntt_release_domain")(), "ntt_release_domain")


def ntt(field: str, inp, direction: int, cfg=None, out=None, size: int = None, extension: bool = False):
    """Scalar-field NTT of a curve (field in SCALAR_NTT_FIELDS): elements are 8 u32 words, cfg is NTTConfigU256."""
    if field == GOLD:
        cfg = cfg or NTTConfigU64.default()
        lanes = 4 if extension else 2  # u32 words per element (quadratic extension: two components)
    elif _is_big(field):
        assert not extension
        cfg = cfg or NTTConfigU256.default()
        lanes = 8
    else:
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


def ecntt(curve: str, inp, direction: int, cfg: NTTConfigU256 = None, out=None, size: int = None):
    """NTT over G1 points (wrappers/rust/icicle-core/src/ecntt/mod.rs): inp is projective_t[size*batch] as uint32
    words (3*L per point, L = 8 bn254 / 12 bls12_381); uses the domain of the curve's scalar-field NTT."""
    from .msm import LIMBS

    cfg = cfg or NTTConfigU256.default()
    ip, i_dev = _ptr(inp)
    cfg.are_inputs_on_device = i_dev
    if size is None:
        size = inp.size // (max(1, cfg.batch_size) * 3 * LIMBS[curve])
    if out is None:
        out = np.zeros_like(inp)
    op, o_dev = _ptr(out)
    cfg.are_outputs_on_device = o_dev
    check(getattr(lib, f"{curve}_ecntt")(ip, size, direction, ctypes.byref(cfg), op), f"{curve}_ecntt")
    return out
