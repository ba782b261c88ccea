#!/usr/bin/env python3
"""Times goldilocks_ntt on device-resident data (no torch: wall clock around stream-synchronised calls)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icicle_amd as H  # noqa: E402
from icicle_amd import ntt as N, runtime  # noqa: E402
from icicle_amd.runtime import DeviceVec  # noqa: E402

runtime.set_device(0)
P = 0xFFFFFFFF00000001
rng = np.random.default_rng(1)
for logn, batch in ((16, 64), (20, 8), (24, 1), (24, 4)):
    n = 1 << logn
    N.init_domain("goldilocks", N.get_root_of_unity("goldilocks", n))
    v = rng.integers(0, 1 << 63, size=n * batch, dtype=np.uint64)
    x = np.ascontiguousarray(v.view(np.uint32))
    d_in, d_out = DeviceVec.from_host(x), DeviceVec.from_host(np.zeros_like(x))
    cfg = H.NTTConfigU64.default()
    cfg.batch_size = batch
    for ext in (False,) if batch * n >= (1 << 26) else (False, True):
        sz = n // 2 if ext else n
        if ext:
            N.release_domain("goldilocks")
            N.init_domain("goldilocks", N.get_root_of_unity("goldilocks", sz))
        run = lambda: N.ntt("goldilocks", d_in, N.FORWARD, cfg, out=d_out, size=sz, extension=ext)
        run()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"ntt goldilocks{'_ext' if ext else '    '} 2^{sz.bit_length() - 1:<2d} batch {batch:<3d} {ms:8.3f} ms  {n * batch / ms / 1e6:7.2f} Gelem/s  {2 * n * batch * 8 / ms / 1e6:6.0f} GB/s algorithmic", flush=True)
    N.release_domain("goldilocks")
