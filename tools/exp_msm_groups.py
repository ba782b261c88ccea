#!/usr/bin/env python3
"""One timing line for a device-resident BN254 MSM under the current environment (the window-group / co-residency switches
are read once per process): exp_msm_groups.py LOGN [REPS] -> 'ms=<median> min=<min> affine=<hash>'. The hash is over the
AFFINE result (the projective representative depends on the order of the additions), via the reference's to_affine."""
import hashlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import msm as M, runtime
from icicle_amd._lib import MSMConfig, lib, check
from oracle import ref

logn = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
curve = os.environ.get("EXP_CURVE", "bn254")
runtime.set_device(0)
dev = torch.device("cuda", 0)
n = 1 << logn
L = M.LIMBS[curve]
bases = torch.empty((n, 2 * L), dtype=torch.int32, device=dev)
check(getattr(lib, f"{curve}_hip_generate_affine_points")(bases.data_ptr(), n, 1, True, None))
g = torch.Generator(device=dev); g.manual_seed(1)
sc = torch.randint(-(2 ** 31), 2 ** 31, (n, 8), dtype=torch.int32, device=dev, generator=g)
sc[:, 7] = torch.randint(0, 0x30644E72 if curve == "bn254" else 0x10000000, (n,), dtype=torch.int32, device=dev, generator=g)
res = torch.empty((1, 3 * L), dtype=torch.int32, device=dev)
cfg = MSMConfig.default(); cfg.is_async = True; cfg.c = int(os.environ.get("EXP_C", "0"))
def run():
    M.msm(curve, sc.data_ptr(), bases.data_ptr(), cfg, results=res.data_ptr(), msm_size=n)
for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
aff = ref.RefCurve(curve).to_affine(res.cpu().numpy().view(np.uint32)) if ref.available(curve) else res.cpu().numpy()
print(f"ms={np.median(ts):.3f} min={min(ts):.3f} affine={hashlib.sha256(aff.tobytes()).hexdigest()[:12]}", flush=True)
