#!/usr/bin/env python3
"""BabyBear NTT 2^LOGN x BATCH in every ordering / direction / coset (device resident): tools/ntt_orderings.py LOGN BATCH"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import ntt as N, runtime  # noqa: E402
from icicle_amd._lib import NTTConfigU32  # noqa: E402

logn, batch = int(sys.argv[1]), int(sys.argv[2])
runtime.set_device(0)
dev = torch.device("cuda", 0)
n = 1 << logn
N.init_domain("babybear", N.get_root_of_unity("babybear", n))
x = torch.randint(0, 0x78000001, (batch, n), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
names = ["kNN", "kNR", "kRN", "kRR", "kNM", "kMN"]
for coset in (1, 3):
    for direction in (N.FORWARD, N.INVERSE):
        for o in range(4):
            cfg = NTTConfigU32.default()
            cfg.batch_size, cfg.is_async, cfg.ordering, cfg.coset_gen = batch, True, o, coset
            f = lambda: N.ntt("babybear", x.data_ptr(), direction, cfg, out=y.data_ptr(), size=n)
            f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            print(f"babybear 2^{logn} x {batch} {names[o]} {'inv' if direction else 'fwd'} coset={coset}: {ms:8.3f} ms", flush=True)
