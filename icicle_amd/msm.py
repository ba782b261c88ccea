"""msm() / precompute_bases(): mirror of wrappers/rust/icicle-core/src/msm/mod.rs:90-207.

Arrays are numpy uint32 in the reference's memory layout: scalars [batch*N, 8]; affine bases
[M, 2*L] (x then y, identity = all zeros); projective results [batch, 3*L] (L = 8 bn254 / grumpkin, 12 bls12_381 /
bls12_377).
With g2=True the group is G2 (not grumpkin): coordinates are Fq2 elements {c0, c1}, so L doubles (16 / 24).
Device-resident operands are passed as runtime.DeviceVec or raw integer device pointers.
"""
import ctypes
import numpy as np
from ._lib import lib, check, MSMConfig
from .runtime import DeviceVec

LIMBS = {"bn254": 8, "bls12_381": 12, "bls12_377": 12, "grumpkin": 8}
SCALAR_LIMBS = 8


def _ptr(x):
    if isinstance(x, DeviceVec):
        return x.ptr, True
    if isinstance(x, int):
        return x, True
    assert isinstance(x, np.ndarray) and x.dtype == np.uint32 and x.flags["C_CONTIGUOUS"], "need contiguous uint32 ndarray"
    return x.ctypes.data, False


def _group(curve: str, g2: bool):
    """(coordinate words, symbol prefix)"""
    return (2 * LIMBS[curve], f"{curve}_g2") if g2 else (LIMBS[curve], curve)


def msm(curve: str, scalars, bases, cfg: MSMConfig = None, results=None, msm_size: int = None, g2: bool = False):
    """results[b] = sum_i scalars[b*N+i] * bases[...]. Returns `results` (host ndarray unless a DeviceVec was given)."""
    L, sym = _group(curve, g2)
    cfg = cfg or MSMConfig.default()
    sp, s_dev = _ptr(scalars)
    bp, b_dev = _ptr(bases)
    cfg.are_scalars_on_device = s_dev
    cfg.are_points_on_device = b_dev
    if msm_size is None:
        assert isinstance(scalars, np.ndarray), "msm_size is required for device scalars"
        total = scalars.size // SCALAR_LIMBS
        assert total % max(1, cfg.batch_size) == 0, "scalars length must be a multiple of batch_size"
        msm_size = total // max(1, cfg.batch_size)
    if results is None:
        results = np.zeros((max(1, cfg.batch_size), 3 * L), dtype=np.uint32)
    rp, r_dev = _ptr(results)
    cfg.are_results_on_device = r_dev
    check(getattr(lib, f"{sym}_msm")(sp, bp, msm_size, ctypes.byref(cfg), rp), f"{sym}_msm")
    return results


def precompute_bases(curve: str, bases, cfg: MSMConfig, output=None, nof_bases: int = None, g2: bool = False):
    L, sym = _group(curve, g2)
    bp, b_dev = _ptr(bases)
    cfg.are_points_on_device = b_dev
    # nof_bases = the bases of ONE MSM, as both wrappers pass it (icicle-core/src/msm/mod.rs:296, golang .../msm/msm.go:39-43):
    # with per-MSM bases (batch_size > 1, are_points_shared_in_batch = False) the input holds nof_bases * batch_size points
    per_msm = max(1, cfg.batch_size) if not cfg.are_points_shared_in_batch else 1
    if nof_bases is None:
        nof_bases = bases.size // (2 * L) // per_msm
    if output is None:
        output = np.zeros((nof_bases * per_msm * cfg.precompute_factor, 2 * L), dtype=np.uint32)
    op, o_dev = _ptr(output)
    cfg.are_results_on_device = o_dev
    check(getattr(lib, f"{sym}_msm_precompute_bases")(bp, nof_bases, ctypes.byref(cfg), op), f"{sym}_msm_precompute_bases")
    return output


def generate_affine_points(curve: str, n: int, k0: int = 1, out=None, g2: bool = False):
    """n distinct points (k0+i)*G generated on the GPU (synthetic benchmark inputs)."""
    L, sym = _group(curve, g2)
    if out is None:
        out = np.zeros((n, 2 * L), dtype=np.uint32)
    op, o_dev = _ptr(out)
    check(getattr(lib, f"{sym}_hip_generate_affine_points")(op, n, k0, o_dev, None), "generate_affine_points")
    return out
