#!/usr/bin/env python3
"""BabyBear forward NTT over a list of (logn, batch) shapes in ONE process -- for same-box A/B rows: run it once per library /
environment variant (ICICLE_HIP_LIB=<path>, ICICLE_HIP_NTT_MIN_BLOCKS=<n>, ...). `ntt_cases.py 24x64 24x8 ...`; a shape
suffixed with c (22x32c) runs the columns_batch layout. Prints ms per direction and a checksum of the output (equal
checksums across variants = same bytes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import ntt as N, runtime  # noqa: E402
from icicle_amd._lib import NTTConfigU32  # noqa: E402

runtime.set_device(0)
dev = torch.device("cuda", 0)
tag = os.environ.get("AB_TAG", "")
for spec in sys.argv[1:]:
    cols = spec.endswith("c")
    logn, batch = (int(v) for v in spec.rstrip("c").split("x"))
    n = 1 << logn
    N.init_domain("babybear", N.get_root_of_unity("babybear", n))
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randint(0, 0x78000001, (batch * n,), dtype=torch.int32, device=dev, generator=g)
    y = torch.empty_like(x)
    cfg = NTTConfigU32.default()
    cfg.batch_size, cfg.is_async, cfg.columns_batch = batch, True, cols
    res = []
    for d in (N.FORWARD, N.INVERSE):
        for _ in range(int(os.environ.get("NTT_CASES_WARM", "3"))):  # (a long warm-up separates clock ramp-up from the kernel's own cost)
            N.ntt("babybear", x.data_ptr(), d, cfg, out=y.data_ptr(), size=n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = int(os.environ.get("NTT_CASES_REPS", "8"))
        for _ in range(reps):
            N.ntt("babybear", x.data_ptr(), d, cfg, out=y.data_ptr(), size=n)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps)
    chk = int(y.to(torch.int64).sum().item()) & 0xFFFFFFFF
    print(f"{tag:22s} babybear 2^{logn} x {batch:<4d}{' columns' if cols else '        '} fwd {res[0]:7.3f} ms  inv {res[1]:7.3f} ms  {2 * batch * n * 4 / res[0] / 1e6:6.0f} GB/s  checksum {chk:08x}", flush=True)
    N.release_domain("babybear")
    del x, y
