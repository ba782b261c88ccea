#!/usr/bin/env python3
"""How the reference CPU MSM scales with its worker count on this host (tests/refpool.py sizes its lanes from this):
tools/ref_scaling.py [logn]  -> one line per n_threads (MSMConfig.ext "n_threads", cpu_msm.hpp:78-100)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
for curve in ("bn254", "bls12_381"):
    refc = ref.RefCurve(curve)
    rng = np.random.default_rng(1)
    bases = refc.generate_affine_points(n)
    sc = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    sc[:, 7] &= 0x0FFFFFFF
    for nt in (8, 16, 32, 64, 128, 0):
        t0 = time.time()
        refc.msm(sc, bases, n_threads=nt)
        dt = time.time() - t0
        print(f"{curve} 2^{logn} n_threads {nt or os.cpu_count():4d}: {dt:7.2f} s  ({dt * (nt or os.cpu_count()):9.0f} core-s)", flush=True)
    if logn > 22:
        break
