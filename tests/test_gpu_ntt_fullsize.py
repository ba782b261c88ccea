"""GPU parity at BASELINE configs[2]'s full size for the NTT path (kept in its own module: it owns the field's twiddle
domain, which test_gpu_ntt.py's module fixture also does). configs[4] runs whole in test_gpu_fullsize_configs.py."""
import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu


def test_babybear_config2_full_size_vs_oracle(hip):
    """BASELINE config 2 itself -- BabyBear 2^24 x 64, device resident -- byte-compared with the reference CPU
    backend (memcmp rule: icicle/tests/test_mod_arithmetic_api.h:694): three rows of the 64-row kNN forward output,
    the inverse of those rows, and one row each of a coset kNR forward and a kRN inverse at the same size (all of
    them 3-pass plans). The CPU side costs one 2^24 domain init + a handful of single-row transforms."""
    import torch
    from icicle_amd import ntt as N

    F = pyref.BABYBEAR
    logn, rows = 24, 64
    n = 1 << logn
    N.init_domain("babybear", N.get_root_of_unity("babybear", n))
    rf = ref.RefNttField("babybear")
    rf.init_domain(rf.get_root_of_unity(n))
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev)
        g.manual_seed(2024)
        x = torch.randint(0, F.p, (rows, n), dtype=torch.int32, device=dev, generator=g)
        y = torch.empty_like(x)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.is_async = rows, True
        N.ntt("babybear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
        torch.cuda.synchronize()
        pick = [0, 37, 63]
        hx = np.ascontiguousarray(x[pick].cpu().numpy().view(np.uint32)).reshape(-1)
        hy = np.ascontiguousarray(y[pick].cpu().numpy().view(np.uint32)).reshape(-1)
        exp = rf.ntt(hx, n, 0, batch=len(pick))
        assert np.array_equal(hy, exp), "forward kNN 2^24 x 64: rows differ from the reference CPU backend"
        # inverse of the full batch, in place on y
        N.ntt("babybear", y.data_ptr(), N.INVERSE, cfg, out=y.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        assert np.array_equal(rf.ntt(exp, n, 1, batch=len(pick)), hx)
        # coset + kNR forward, kRN inverse on a 4-row batch of the same size; row 2 against the oracle
        cfg4 = hip.NTTConfigU32.default()
        cfg4.batch_size, cfg4.is_async = 4, True
        cfg4.coset_gen = 31
        cfg4.ordering = N.kNR
        y4 = torch.empty((4, n), dtype=torch.int32, device=dev)
        N.ntt("babybear", x[:4].data_ptr(), N.FORWARD, cfg4, out=y4.data_ptr(), size=n)
        torch.cuda.synchronize()
        row = np.ascontiguousarray(x[2].cpu().numpy().view(np.uint32))
        e = rf.ntt(row, n, 0, ordering=N.kNR, coset_gen=31)
        assert np.array_equal(y4[2].cpu().numpy().view(np.uint32), e), "coset kNR forward 2^24"
        cfg4.ordering = N.kRN
        z4 = torch.empty_like(y4)
        N.ntt("babybear", y4.data_ptr(), N.INVERSE, cfg4, out=z4.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert np.array_equal(z4[2].cpu().numpy().view(np.uint32), rf.ntt(e, n, 1, ordering=N.kRN, coset_gen=31))
        assert torch.equal(z4, x[:4])
    finally:
        N.release_domain("babybear")
        rf.release_domain()
