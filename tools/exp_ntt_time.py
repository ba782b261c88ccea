#!/usr/bin/env python3
"""BabyBear NTT timing for same-box A/B runs: exp_ntt_time.py [LOGN BATCH] -> ms per direction (forward, then inverse) + round trip check"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import ntt as N, runtime
from icicle_amd._lib import NTTConfigU32
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
runtime.set_device(0)
dev = torch.device("cuda", 0)
n = 1 << logn
N.init_domain("babybear", N.get_root_of_unity("babybear", n))
pad_mb = int(os.environ.get("EXP_PAD_MB", "0"))  # shift the buffers' placement: allocate this much first
pad = torch.empty((pad_mb << 20,), dtype=torch.uint8, device=dev) if pad_mb else None
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randint(0, 0x78000001, (batch, n), dtype=torch.int32, device=dev, generator=g)
y, z = torch.empty_like(x), torch.empty_like(x)
cfg = NTTConfigU32.default(); cfg.batch_size = batch; cfg.is_async = True
def run(direction, a, b, reps=6):
    N.ntt("babybear", a.data_ptr(), direction, cfg, out=b.data_ptr(), size=n); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        N.ntt("babybear", a.data_ptr(), direction, cfg, out=b.data_ptr(), size=n)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
if os.environ.get("EXP_PREHEAT"):  # an ALU-bound phase right before (what bench.py's MSM section is to its NTT section)
    import ctypes
    from icicle_amd._lib import lib
    r = ctypes.c_double()
    for _ in range(int(os.environ["EXP_PREHEAT"])):
        lib.icicle_hip_ubench_mixed_add(0, ctypes.byref(r))
if os.environ.get("EXP_SLEEP"):
    time.sleep(float(os.environ["EXP_SLEEP"]))
if os.environ.get("EXP_TIMER"):  # the hipEvent kernel timer bench.py switches on
    from icicle_amd._lib import lib
    lib.icicle_hip_enable_kernel_timing(True)
if os.environ.get("EXP_ALT"):  # forward / inverse alternating, as bench.py's ntt_step does
    def both(reps=3):
        N.ntt("babybear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
        N.ntt("babybear", y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            N.ntt("babybear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
            N.ntt("babybear", y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=n)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps / 2
    print(f"alternating: {both():.3f} ms per direction", flush=True)
f = run(N.FORWARD, x, y); i = run(N.INVERSE, y, z)
print(f"babybear 2^{logn} x {batch}: forward {f:.3f} ms  inverse {i:.3f} ms  roundtrip_ok={bool(torch.equal(x, z))}  checksum={int(y.to(torch.int64).sum().item()) & 0xffffffff:08x}", flush=True)
N.release_domain("babybear")
