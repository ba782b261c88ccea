import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from icicle_amd import msm as M, ntt as N, runtime
from icicle_amd._lib import NTTConfigU256, lib, check
runtime.set_device(0); dev = torch.device("cuda", 0)
def t(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05: fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
for logn in (10, 12, 14):
    n = 1 << logn
    N.init_domain("bn254", N.get_root_of_unity("bn254", n))
    aff = torch.empty((n, 16), dtype=torch.int32, device=dev)
    check(lib.bn254_hip_generate_affine_points(aff.data_ptr(), n, 7, True, None))
    one = torch.zeros((n, 8), dtype=torch.int32, device=dev); one[:, 0] = 1
    x = torch.cat([aff, one], dim=1).contiguous(); y = torch.empty_like(x)
    row = []
    for d, cos in ((N.FORWARD, 1), (N.INVERSE, 1), (N.FORWARD, 5), (N.INVERSE, 5)):
        cfg = NTTConfigU256.default(); cfg.is_async = True; cfg.set_coset_gen(cos)
        row.append(f"{'fwd' if d == N.FORWARD else 'inv'}{' coset' if cos != 1 else ''} {t(lambda: N.ecntt('bn254', x.data_ptr(), d, cfg, out=y.data_ptr(), size=n)):.2f}")
    print(f"ecntt bn254 2^{logn}: " + ", ".join(row) + " ms", flush=True)
    N.release_domain("bn254")
