#!/usr/bin/env python3
"""One NTT case on the GPU (profiling helper): tools/ntt_one.py LOGN BATCH [reps]   (NTT_ONE_COLUMNS=1: columns_batch,
NTT_ONE_ORDERING=<0..5>, NTT_ONE_FIELD=babybear|koalabear, NTT_ONE_ROUNDTRIP=1: forward + inverse per repetition)"""
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
field = os.environ.get("NTT_ONE_FIELD", "babybear")
P = {"babybear": 0x78000001, "koalabear": 0x7F000001}[field]
N.init_domain(field, N.get_root_of_unity(field, n))
x = torch.randint(0, P, (batch, n), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
cfg = NTTConfigU32.default()
cfg.batch_size = batch
cfg.is_async = True
cfg.columns_batch = os.environ.get("NTT_ONE_COLUMNS", "0") == "1"
cfg.ordering = int(os.environ.get("NTT_ONE_ORDERING", "0"))
roundtrip = os.environ.get("NTT_ONE_ROUNDTRIP", "0") == "1"
z = torch.empty_like(x) if roundtrip else None
for _ in range(reps):
    N.ntt(field, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
    if roundtrip:
        N.ntt(field, y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=n)
torch.cuda.synchronize()
