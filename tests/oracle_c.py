"""ctypes access to oracle/_build/liboracle.so (the plain-C restatement). Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
CURVE_ID = {"bn254": 0, "bls12_381": 1, "bls12_377": 2, "grumpkin": 3}
LIMBS = {"bn254": 8, "bls12_381": 12, "bls12_377": 12, "grumpkin": 8}
FIELD_ID = {"babybear": 0, "koalabear": 1}
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "oracle.c")
        if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _lib = ctypes.CDLL(SO)
        _lib.oracle_omega.restype = ctypes.c_uint32
    return _lib


def msm(curve, scalars, bases, c=8, bitsize=0):
    L = LIMBS[curve]
    n = scalars.size // 8
    out = np.zeros(3 * L, dtype=np.uint32)
    rc = lib().oracle_msm(CURVE_ID[curve], scalars.ctypes.data_as(ctypes.c_void_p), bases.ctypes.data_as(ctypes.c_void_p),
                          n, c, bitsize, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def to_affine(curve, proj):
    L = LIMBS[curve]
    proj = np.ascontiguousarray(proj.reshape(3 * L))
    out = np.zeros(2 * L, dtype=np.uint32)
    assert lib().oracle_to_affine(CURVE_ID[curve], proj.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
    return out


def ntt(field, x, n, domain_root, inverse=False, ordering=0, coset_gen=1, batch=1, columns_batch=False, lanes=1):
    x = np.ascontiguousarray(x)
    out = np.zeros_like(x)
    rc = lib().oracle_ntt(FIELD_ID[field], x.ctypes.data_as(ctypes.c_void_p), n, ctypes.c_uint32(domain_root), int(inverse),
                          ordering, ctypes.c_uint32(coset_gen), batch, int(columns_batch), lanes,
                          out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out


def omega(field, logn):
    return lib().oracle_omega(FIELD_ID[field], logn)
