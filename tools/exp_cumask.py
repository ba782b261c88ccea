#!/usr/bin/env python3
"""Experiment: how do the phases of one BN254 2^26 MSM scale with the number of CUs?

The sort in front of bucket accumulation (7.7 ms) and the reduction behind it (3.3 ms) are latency / memory-system
bound, the accumulation is VALU-issue bound. Overlapping them on two streams only pays if the sort does NOT need the
whole chip: then it can run on a few CUs while accumulation keeps the rest. A stream created with
hipExtStreamCreateWithCUMask restricts every kernel launched on it to the CUs of the mask (bits are dealt round-robin
over the 8 XCDs), so the phase timers of the library (icicle_hip_kernel_timing 2 / 0 / 3) give T_phase(k CUs) directly.
usage: python tools/exp_cumask.py [size_log2=26]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
import icicle_amd  # noqa: E402
from icicle_amd import msm as M, runtime  # noqa: E402
from icicle_amd._lib import MSMConfig, lib, check  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << logn
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
runtime.set_device(0)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, 1, True, None))
g = torch.Generator(device=dev)
g.manual_seed(1)
sc = torch.randint(-(2 ** 31), 2 ** 31, (n, 8), dtype=torch.int32, device=dev, generator=g)
sc[:, 7] = torch.randint(0, 0x30644E72, (n,), dtype=torch.int32, device=dev, generator=g)
out = torch.empty(24, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
tot, cnt = ctypes.c_double(), ctypes.c_int()
print(f"BN254 MSM 2^{logn}: phase time (ms) vs CUs enabled on the launch stream")
print(f"{'CUs':>5} {'whole':>8} {'sort':>8} {'accum':>8} {'reduce':>8}")
for k in (256, 224, 192, 128, 96, 64, 32):
    words = (ctypes.c_uint32 * 8)(*[((1 << min(32, max(0, k - 32 * i))) - 1) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        print(f"{k:5d} hipExtStreamCreateWithCUMask failed rc={rc}")
        continue
    cfg = MSMConfig.default()
    cfg.stream = st.value
    M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfg, results=out.data_ptr(), msm_size=n)  # warm
    lib.icicle_hip_enable_kernel_timing(True)
    for w in (0, 2, 3):
        lib.icicle_hip_kernel_timing(w, True, ctypes.byref(tot), ctypes.byref(cnt))
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfg, results=out.data_ptr(), msm_size=n)
    dt = (time.perf_counter() - t0) / reps * 1e3
    ph = {}
    for w in (2, 0, 3):
        lib.icicle_hip_kernel_timing(w, True, ctypes.byref(tot), ctypes.byref(cnt))
        ph[w] = tot.value / max(1, cnt.value)
    lib.icicle_hip_enable_kernel_timing(False)
    print(f"{k:5d} {dt:8.2f} {ph[2]:8.2f} {ph[0]:8.2f} {ph[3]:8.2f}", flush=True)
