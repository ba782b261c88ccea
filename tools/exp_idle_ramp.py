"""Per-call time of the headline NTT and MSM right after an idle gap of 0 / 0.3 / 2 s (clock ramp): exp_idle_ramp.py"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icicle_amd import msm as M, ntt as N, runtime
from icicle_amd._lib import MSMConfig, NTTConfigU32, lib, check
runtime.set_device(0); dev = torch.device("cuda", 0)
logn, rows = 24, 64; nn = 1 << logn
N.init_domain("babybear", N.get_root_of_unity("babybear", nn))
x = torch.randint(0, 0x78000001, (rows, nn), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
cfg = NTTConfigU32.default(); cfg.batch_size = rows; cfg.is_async = True
def ntt(): N.ntt("babybear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=nn)
ntt(); torch.cuda.synchronize()
for gap in (0.0, 0.3, 2.0):
    time.sleep(gap)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for i in range(60):
        ntt(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(60)]
    print(f"ntt after {gap:.1f} s idle: " + " ".join(f"{v:.2f}" for v in t[:12]) + " ... " + " ".join(f"{t[i]:.2f}" for i in (19, 29, 39, 49, 59)), flush=True)
n = 1 << 26
bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, 1, True, None))
sc = torch.randint(-(2 ** 31), 2 ** 31, (n, 8), dtype=torch.int32, device=dev); sc[:, 7] = torch.randint(0, 0x30644E72, (n,), dtype=torch.int32, device=dev)
res = torch.empty((1, 24), dtype=torch.int32, device=dev)
mc = MSMConfig.default(); mc.is_async = True
def msm(): M.msm("bn254", sc.data_ptr(), bases.data_ptr(), mc, results=res.data_ptr(), msm_size=n)
msm(); torch.cuda.synchronize()
for gap in (0.0, 0.3, 2.0):
    time.sleep(gap)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ev[0].record()
    for i in range(8):
        msm(); ev[i + 1].record()
    torch.cuda.synchronize()
    print(f"msm after {gap:.1f} s idle: " + " ".join(f"{ev[i].elapsed_time(ev[i+1]):.2f}" for i in range(8)), flush=True)
