#!/usr/bin/env python3
"""Experiment: shader clock and power while bucket accumulation runs on all CUs vs on half of them (CU-masked stream).
profiles/r03_notes.md section 1 infers from timing that the all-v_mad_u64_u32 stream is power-limited at full width;
this samples rocm-smi while a BN254 2^24 MSM loops. usage: python tools/exp_clock.py"""
import ctypes
import json
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from icicle_amd import msm as M, runtime  # noqa: E402
from icicle_amd._lib import MSMConfig, lib, check  # noqa: E402

n = 1 << 24
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


def sample(stop, rows):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            rows.append({k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception as e:  # noqa: BLE001
            rows.append({"error": repr(e)})
        time.sleep(0.1)


for k in (256, 128, 64):
    words = (ctypes.c_uint32 * 8)(*[((1 << min(32, max(0, k - 32 * i))) - 1) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    cfg = MSMConfig.default()
    cfg.stream = st.value
    M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfg, results=out.data_ptr(), msm_size=n)
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, rows))
    th.start()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 4.0:
        M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfg, results=out.data_ptr(), msm_size=n)
        reps += 1
    dt = (time.perf_counter() - t0) / reps * 1e3
    stop.set()
    th.join()
    print(f"--- {k} CUs: {dt:.2f} ms per 2^24 MSM, {len(rows)} samples")
    for r in rows[1:-1][:6]:
        print("   ", r)
