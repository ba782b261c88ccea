#!/usr/bin/env python3
"""k_accumulate time per mixed add as a function of the bases footprint: exp_acc_time.py LOGN:C [LOGN:C ...]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import msm as M, runtime
from icicle_amd._lib import MSMConfig, lib, check
runtime.set_device(0)
dev = torch.device("cuda", 0)
for spec in sys.argv[1:]:
    logn, c = (int(v) for v in spec.split(":"))
    n = 1 << logn
    bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
    check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, 1, True, None))
    g = torch.Generator(device=dev); g.manual_seed(1)
    sc = torch.randint(-(2 ** 31), 2 ** 31, (n, 8), dtype=torch.int32, device=dev, generator=g)
    sc[:, 7] = torch.randint(0, 0x30644E72, (n,), dtype=torch.int32, device=dev, generator=g)
    res = torch.empty((1, 24), dtype=torch.int32, device=dev)
    cfg = MSMConfig.default(); cfg.is_async = True; cfg.c = c
    run = lambda: M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfg, results=res.data_ptr(), msm_size=n)
    run(); torch.cuda.synchronize()
    tot, cnt = ctypes.c_double(), ctypes.c_int()
    lib.icicle_hip_enable_kernel_timing(True)
    lib.icicle_hip_kernel_timing(0, True, ctypes.byref(tot), ctypes.byref(cnt))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    lib.icicle_hip_kernel_timing(0, True, ctypes.byref(tot), ctypes.byref(cnt))
    lib.icicle_hip_enable_kernel_timing(False)
    pc, pw = ctypes.c_int(), ctypes.c_int()
    check(lib.icicle_hip_msm_plan(n, 254, ctypes.byref(cfg), ctypes.byref(pc), ctypes.byref(pw)))
    acc = tot.value / max(1, cnt.value)
    adds = n * pw.value
    print(f"2^{logn} c={pc.value} windows={pw.value}: msm {e0.elapsed_time(e1)/3:8.3f} ms  accumulate {acc:8.3f} ms  {acc*1e9/adds:6.1f} ps/add  {adds/acc/1e6:6.2f} Gadd/s", flush=True)
    del bases, sc
