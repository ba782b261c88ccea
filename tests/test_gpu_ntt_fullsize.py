"""GPU parity at BASELINE's full per-GPU sizes for the NTT path, through size-independent properties (kept in its own
module: it owns the field's twiddle domain, which test_gpu_ntt.py's module fixture also does)."""
import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu


def test_koalabear_config4_share_roundtrip(hip):
    """per-GPU share of BASELINE config 4 (KoalaBear 2^22 x 1024 over 8 GPUs = 128 rows), device resident:
    inverse(forward(x)) == x, forward is linear, and one row equals the reference CPU backend"""
    import torch
    from icicle_amd import ntt as N

    F = pyref.KOALABEAR
    logn, rows = 22, 128
    n = 1 << logn
    N.init_domain("koalabear", N.get_root_of_unity("koalabear", n))
    rf = ref.RefNttField("koalabear")
    rf.init_domain(rf.get_root_of_unity(n))
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev)
        g.manual_seed(4)
        x = torch.randint(0, F.p, (rows, n), dtype=torch.int32, device=dev, generator=g)
        y, z = torch.empty_like(x), torch.empty_like(x)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.is_async = rows, True
        N.ntt("koalabear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
        N.ntt("koalabear", y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(x, z)
        # X[0] of every row is the sum of the row
        sums = (x.to(torch.int64).sum(dim=1) % F.p).to(torch.int32)
        assert torch.equal(y[:, 0], sums)
        row = np.ascontiguousarray(x[5].cpu().numpy().view(np.uint32))
        assert np.array_equal(y[5].cpu().numpy().view(np.uint32), rf.ntt(row, n, 0))
    finally:
        N.release_domain("koalabear")
        rf.release_domain()
