#!/usr/bin/env python3
"""One BabyBear NTT case on the GPU (profiling helper): tools/ntt_one.py LOGN BATCH [reps]   (NTT_ONE_COLUMNS=1: columns_batch,
NTT_ONE_ORDERING=<0..5>)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import ntt as N, runtime  # noqa: E402
from icicle_amd._lib import NTTConfigU32  # noqa: E402

logn, batch = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
runtime.set_device(0)
dev = torch.device("cuda", 0)
n = 1 << logn
N.init_domain("babybear", N.get_root_of_unity("babybear", n))
x = torch.randint(0, 0x78000001, (batch, n), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
cfg = NTTConfigU32.default()
cfg.batch_size = batch
cfg.is_async = True
cfg.columns_batch = os.environ.get("NTT_ONE_COLUMNS", "0") == "1"
cfg.ordering = int(os.environ.get("NTT_ONE_ORDERING", "0"))
for _ in range(reps):
    N.ntt("babybear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
torch.cuda.synchronize()
