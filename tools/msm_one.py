#!/usr/bin/env python3
"""Run a few device-resident MSMs of one size (for rocprofv3 kernel traces): msm_one.py CURVE LOGN [BATCH] [C]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import msm as M, runtime
from icicle_amd._lib import MSMConfig, lib, check
curve, logn = sys.argv[1], int(sys.argv[2])
g2 = curve.endswith("_g2")
if g2:
    curve = curve[:-3]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = int(sys.argv[4]) if len(sys.argv) > 4 else 0
runtime.set_device(0)
dev = torch.device("cuda", 0)
n = 1 << logn
L = M.LIMBS[curve] * (2 if g2 else 1)
sym = f"{curve}_g2" if g2 else curve
bases = torch.empty((n, 2 * L), dtype=torch.int32, device=dev)
check(getattr(lib, f"{sym}_hip_generate_affine_points")(bases.data_ptr(), n, 1, True, None))
g = torch.Generator(device=dev); g.manual_seed(1)
sc = torch.randint(-(2 ** 31), 2 ** 31, (n * batch, 8), dtype=torch.int32, device=dev, generator=g)
sc[:, 7] = torch.randint(0, 0x30644E72, (n * batch,), dtype=torch.int32, device=dev, generator=g)
res = torch.empty((batch, 3 * L), dtype=torch.int32, device=dev)
cfg = MSMConfig.default(); cfg.batch_size = batch; cfg.is_async = True; cfg.c = c
for _ in range(3):
    M.msm(curve, sc.data_ptr(), bases.data_ptr(), cfg, results=res.data_ptr(), msm_size=n, g2=g2)
torch.cuda.synchronize()
