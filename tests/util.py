"""Shared helpers for the parity tests (host-side only)."""
import numpy as np

from oracle import pyref


def to_words(vals, nlimbs):
    out = np.zeros((len(vals), nlimbs), dtype=np.uint32)
    for i, v in enumerate(vals):
        for k in range(nlimbs):
            out[i, k] = (v >> (32 * k)) & 0xFFFFFFFF
    return out


def from_words(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return sum(int(x) << (32 * k) for k, x in enumerate(a))
    return [sum(int(x) << (32 * k) for k, x in enumerate(row)) for row in a]


def rand_scalars(rng, n, modulus, bits=None):
    """n x 8 uint32, uniform below `modulus` (or below 2^bits)."""
    out = []
    for _ in range(n):
        v = int.from_bytes(rng.bytes(40), "little")
        out.append(v % (1 << bits) if bits else v % modulus)
    return out


def points_to_array(curve: pyref.Curve, pts):
    L = curve.limbs_q
    return np.concatenate([to_words([p[0] for p in pts], L), to_words([p[1] for p in pts], L)], axis=1)


def proj_to_affine_py(curve: pyref.Curve, proj_row):
    L = curve.limbs_q
    x, y, z = (from_words(proj_row[i * L:(i + 1) * L]) for i in range(3))
    return pyref.proj_to_affine(curve, x, y, z), (x, y, z)


_POINT_CACHE = {}


def cached_points(curve: pyref.Curve, n, k0=987654321):
    """n distinct points (k0+i)G, generated once per session in pure Python (~20 us/point)."""
    key = (curve.name, k0)
    have = _POINT_CACHE.get(key, [])
    if len(have) < n:
        have = pyref.gen_points(curve, n, k0)
        _POINT_CACHE[key] = have
    return have[:n]


def combine_partials_host(curve: str, partials: np.ndarray):
    """Host-side DEFINITION of the multi-GPU combine step (sum of projective partial results) with the
    pure-Python oracle; the product does this with k_proj_sum on the GPU. partials: [world, 3*L] uint32."""
    C = pyref.CURVES[curve]
    L = C.limbs_q
    acc = pyref.INF
    for row in partials:
        x, y, z = (sum(int(v) << (32 * k) for k, v in enumerate(row[i * L:(i + 1) * L])) for i in range(3))
        acc = pyref.ec_add(C, acc, pyref.proj_to_affine(C, x, y, z))
    return acc
