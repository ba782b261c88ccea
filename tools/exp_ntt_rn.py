#!/usr/bin/env python3
"""kNN vs kNR vs kRN (device resident, warm), rows and columns_batch, + the polynomial-product pipeline
kNR forward -> vector_mul -> kRN inverse: tools/exp_ntt_rn.py  (ICICLE_HIP_NTT_RN_NATIVE=0: rounds 1-4's reordering pre-pass)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import ntt as N, runtime, vecops as V
from icicle_amd._lib import NTTConfigU32

runtime.set_device(0)
dev = torch.device("cuda", 0)
tag = "native" if os.environ.get("ICICLE_HIP_NTT_RN_NATIVE", "1") != "0" else "prepass"

def med(f, reps=9):
    for _ in range(4):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

for field, p in (("babybear", 0x78000001), ("koalabear", 0x7F000001)):
    for logn, batch in ((24, 64), (22, 64), (20, 100), (16, 1024)) if field == "babybear" else ((22, 64),):
        n = 1 << logn
        N.init_domain(field, N.get_root_of_unity(field, n))
        x = torch.randint(0, p, (batch * n,), dtype=torch.int32, device=dev)
        y = torch.empty_like(x)
        for layout in ("rows", "columns"):
            row = []
            for name, o, d, coset in (("kNN fwd", N.kNN, N.FORWARD, 1), ("kNR fwd", N.kNR, N.FORWARD, 1), ("kRN inv", N.kRN, N.INVERSE, 1), ("kRN fwd", N.kRN, N.FORWARD, 1), ("kRN inv coset", N.kRN, N.INVERSE, 5), ("kRR fwd", N.kRR, N.FORWARD, 1)):
                cfg = NTTConfigU32.default()
                cfg.batch_size, cfg.is_async, cfg.ordering, cfg.columns_batch, cfg.coset_gen = batch, True, o, layout == "columns", coset
                ms = med(lambda: N.ntt(field, x.data_ptr(), d, cfg, out=y.data_ptr(), size=n))
                row.append(f"{name} {ms:7.3f}")
            print(f"[{tag}] {field} 2^{logn} x {batch} {layout:7s}: " + " | ".join(row) + " ms", flush=True)
        N.release_domain(field)
