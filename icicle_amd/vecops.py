"""Montgomery-form conversion on device: mirror of wrappers/rust/icicle-core/src/vec_ops (convert_montgomery)
and icicle-core/src/curve (Affine/Projective to_mont / from_mont)."""
import ctypes
import numpy as np
from ._lib import lib, check, VecOpsConfig
from .runtime import DeviceVec

# u32 words per scalar (a curve name stands for its scalar field) and per base-field coordinate of a point
_WORDS = {"bn254": 8, "bls12_381": 8, "bls12_377": 8, "grumpkin": 8, "stark252": 8, "goldilocks": 2, "babybear": 1, "koalabear": 1}
_EXT_DEGREE = {"babybear": 4, "koalabear": 4, "goldilocks": 2}
_POINT_LIMBS = {"bn254": 8, "bls12_381": 12, "bls12_377": 12, "grumpkin": 8}

def _ptr(x):
    if isinstance(x, DeviceVec):
        return x.ptr, True
    if isinstance(x, int):
        return x, True
    assert isinstance(x, np.ndarray) and x.dtype == np.uint32 and x.flags["C_CONTIGUOUS"]
    return x.ctypes.data, False


def _run(symbol, inp, count, to_montgomery, cfg, out):
    cfg = cfg or VecOpsConfig.default()
    ip, i_dev = _ptr(inp)
    cfg.is_a_on_device = i_dev
    if out is None:
        out = np.zeros_like(inp)
    op, o_dev = _ptr(out)
    cfg.is_result_on_device = o_dev
    check(getattr(lib, symbol)(ip, count, to_montgomery, ctypes.byref(cfg), op), symbol)
    return out


def scalar_convert_montgomery(field: str, inp, to_montgomery: bool, cfg=None, out=None, size=None, extension=False):
    """field in _WORDS (a curve name means its scalar field); `size` = elements per batch entry"""
    words = _WORDS[field] * (_EXT_DEGREE[field] if extension else 1)
    if size is None:
        size = inp.size // words // max(1, (cfg.batch_size if cfg else 1))
    sym = f"{field}_extension_scalar_convert_montgomery" if extension else f"{field}_scalar_convert_montgomery"
    return _run(sym, inp, size, to_montgomery, cfg, out)


def affine_convert_montgomery(curve: str, inp, to_montgomery: bool, cfg=None, out=None, n=None):
    L = _POINT_LIMBS[curve]
    return _run(f"{curve}_affine_convert_montgomery", inp, n if n is not None else inp.size // (2 * L), to_montgomery, cfg, out)


def projective_convert_montgomery(curve: str, inp, to_montgomery: bool, cfg=None, out=None, n=None):
    L = _POINT_LIMBS[curve]
    return _run(f"{curve}_projective_convert_montgomery", inp, n if n is not None else inp.size // (3 * L), to_montgomery, cfg, out)


def _vec2(op: str, field: str, a, b, size, cfg, out):
    """mirror of wrappers/rust/icicle-core/src/vec_ops: add / sub / mul / scalar-mul of field vectors"""
    cfg = cfg or VecOpsConfig.default()
    ap, cfg.is_a_on_device = _ptr(a)
    bp, cfg.is_b_on_device = _ptr(b)
    if size is None:
        size = b.size // _WORDS[field] // max(1, cfg.batch_size)
    if out is None:
        out = np.zeros_like(b)
    op_, cfg.is_result_on_device = _ptr(out)
    check(getattr(lib, f"{field}_{op}")(ap, bp, size, ctypes.byref(cfg), op_), f"{field}_{op}")
    return out


def vector_add(field, a, b, cfg=None, out=None, size=None):
    return _vec2("vector_add", field, a, b, size, cfg, out)


def vector_sub(field, a, b, cfg=None, out=None, size=None):
    return _vec2("vector_sub", field, a, b, size, cfg, out)


def vector_mul(field, a, b, cfg=None, out=None, size=None):
    return _vec2("vector_mul", field, a, b, size, cfg, out)


def scalar_mul_vec(field, scalars, b, cfg=None, out=None, size=None):
    """scalars: one per batch entry"""
    return _vec2("scalar_mul_vec", field, scalars, b, size, cfg, out)


def bit_reverse(field, inp, cfg=None, out=None, size=None):
    cfg = cfg or VecOpsConfig.default()
    ip, cfg.is_a_on_device = _ptr(inp)
    if size is None:
        size = inp.size // _WORDS[field] // max(1, cfg.batch_size)
    if out is None:
        out = np.zeros_like(inp)
    op_, cfg.is_result_on_device = _ptr(out)
    check(getattr(lib, f"{field}_bit_reverse")(ip, size, ctypes.byref(cfg), op_), f"{field}_bit_reverse")
    return out


def matrix_transpose(field, inp, nof_rows: int, nof_cols: int, cfg=None, out=None, extension: bool = False):
    """batch_size row-major nof_rows x nof_cols matrices -> their transposes (icicle/include/icicle/vec_ops.h:319,
    src/matrix_ops.cpp:75-102); in place allowed, columns_batch rejected like the reference CPU backend does"""
    cfg = cfg or VecOpsConfig.default()
    ip, cfg.is_a_on_device = _ptr(inp)
    if out is None:
        out = np.zeros_like(inp)
    op_, cfg.is_result_on_device = _ptr(out)
    name = f"{field}_extension_matrix_transpose" if extension else f"{field}_matrix_transpose"
    check(getattr(lib, name)(ip, nof_rows, nof_cols, ctypes.byref(cfg), op_), name)
    return out
