#!/usr/bin/env python3
"""Mid-size BN254 MSMs, device-resident operands, result on the HOST (the wrappers' default) against result on the device:
exp_msm_midsize.py [logn ...]. Run once per ICICLE_HIP_MSM_HOST_COMBINE setting (read once per process)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import msm as M, runtime
from icicle_amd._lib import MSMConfig, lib, check

runtime.set_device(0)
dev = torch.device("cuda", 0)
logns = [int(a) for a in sys.argv[1:]] or [12, 14, 16, 18, 20, 21, 22]
nmax = 1 << max(logns)
bases = torch.empty((nmax, 16), dtype=torch.int32, device=dev)
check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), nmax, 1, True, None))
g = torch.Generator(device=dev); g.manual_seed(1)
sc = torch.randint(-(2 ** 31), 2 ** 31, (nmax, 8), dtype=torch.int32, device=dev, generator=g)
sc[:, 7] = torch.randint(0, 0x30644E72, (nmax,), dtype=torch.int32, device=dev, generator=g)
dres = torch.empty((1, 24), dtype=torch.int32, device=dev)
hres = np.zeros((1, 24), dtype=np.uint32)

def med(fn, reps=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

# keep the clocks up: a long MSM first
cfgw = MSMConfig.default(); cfgw.is_async = True
for _ in range(3):
    M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfgw, results=dres.data_ptr(), msm_size=nmax)
for logn in logns:
    n = 1 << logn
    cd = MSMConfig.default(); cd.is_async = True
    ch = MSMConfig.default()
    td = med(lambda: M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cd, results=dres.data_ptr(), msm_size=n))
    th = med(lambda: M.msm("bn254", sc.data_ptr(), bases.data_ptr(), ch, results=hres, msm_size=n))
    same = np.array_equal(hres, dres.cpu().numpy().view(np.uint32))
    print(f"bn254 2^{logn:<2d} result on device {td:7.3f} ms   result on host {th:7.3f} ms   host_combine={os.environ.get('ICICLE_HIP_MSM_HOST_COMBINE', '1')}  same_words={same}", flush=True)
