"""GPU parity at 3-pass sizes for the two INTERLEAVED layouts of the 31-bit NTT:

* `columns_batch` -- element j of transform b at j * batch + b (icicle/backend/cpu/include/ntt_cpu.h:250,274-275; the layout
  the Rust suite produces with `matrix_transpose`, wrappers/rust/icicle-core/src/ntt/tests.rs:300-340);
* the quartic extension field -- four base-field transforms per row (icicle/src/ntt.cpp:88-101).

Both run on lane-native tiles (icicle_amd/csrc/ntt_fast.hpp, LN): whole columns / rows are memcmp'd with the reference CPU
backend (oracle/_ref), the rest is pinned through the round trip and through the equivalence with the row-major transform
of the transposed matrix, which is itself oracle-checked at full size in test_gpu_ntt_fullsize.py."""
import numpy as np
import pytest

from oracle import pyref, ref

pytestmark = pytest.mark.gpu


def _domain(N, fname, n):
    N.init_domain(fname, N.get_root_of_unity(fname, n))
    rf = ref.RefNttField(fname)
    rf.init_domain(rf.get_root_of_unity(n))
    return rf


def test_columns_batch_2_22_x_64_vs_oracle(hip):
    """KoalaBear, 64 interleaved transforms of 2^22 points (1 GiB): forward + inverse, kNN; four columns against the
    reference, every column against the row-major transform of the transposed matrix; then coset + kNR forward and the
    kRN inverse back (the reordering pre-pass + the bit-reversed store of the lane-native tiles)."""
    import torch
    from icicle_amd import ntt as N

    fname, F = "koalabear", pyref.KOALABEAR
    logn, B = 22, 64
    n = 1 << logn
    rf = _domain(N, fname, n)
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev)
        g.manual_seed(404)
        x = torch.randint(0, F.p, (n, B), dtype=torch.int32, device=dev, generator=g)  # [element][transform]
        y = torch.empty_like(x)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.columns_batch, cfg.is_async = B, True, True
        N.ntt(fname, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
        torch.cuda.synchronize()
        pick = [0, 31, 32, 63]
        hx = np.ascontiguousarray(x[:, pick].t().cpu().numpy().view(np.uint32)).reshape(-1)
        hy = np.ascontiguousarray(y[:, pick].t().cpu().numpy().view(np.uint32)).reshape(-1)
        exp = rf.ntt(hx, n, 0, batch=len(pick))
        assert np.array_equal(hy, exp), "columns_batch forward 2^22 x 64: columns differ from the reference CPU backend"
        # all 64 columns: the same transforms as rows of the transposed matrix
        xt = x.t().contiguous()
        yt = torch.empty_like(xt)
        cfgr = hip.NTTConfigU32.default()
        cfgr.batch_size, cfgr.is_async = B, True
        N.ntt(fname, xt.data_ptr(), N.FORWARD, cfgr, out=yt.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(yt.t(), y)
        del xt, yt
        # inverse, in place
        N.ntt(fname, y.data_ptr(), N.INVERSE, cfg, out=y.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        assert np.array_equal(rf.ntt(exp, n, 1, batch=len(pick)), hx)
        # coset + kNR forward; column 5 and column 40 against the reference; kRN inverse returns the input
        cfg.coset_gen, cfg.ordering = 31, N.kNR
        N.ntt(fname, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n)
        torch.cuda.synchronize()
        for col in (5, 40):
            c_in = np.ascontiguousarray(x[:, col].cpu().numpy().view(np.uint32))
            e = rf.ntt(c_in, n, 0, ordering=N.kNR, coset_gen=31)
            assert np.array_equal(y[:, col].cpu().numpy().view(np.uint32), e), f"coset kNR forward, column {col}"
        cfg.ordering = N.kRN
        z = torch.empty_like(x)
        N.ntt(fname, y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(z, x)
    finally:
        N.release_domain(fname)
        rf.release_domain()


def test_extension_ntt_2_22_x_8_vs_oracle(hip):
    """BabyBear quartic extension, 8 rows of 2^22 extension elements (512 MiB): forward + inverse; two whole rows
    against the reference's extension_ntt, all rows against four base-field transforms of the de-interleaved lanes;
    coset + kNR on one row."""
    import torch
    from icicle_amd import ntt as N

    fname, F = "babybear", pyref.BABYBEAR
    logn, rows = 22, 8
    n = 1 << logn
    rf = _domain(N, fname, n)
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev)
        g.manual_seed(808)
        x = torch.randint(0, F.p, (rows, n, 4), dtype=torch.int32, device=dev, generator=g)
        y = torch.empty_like(x)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size, cfg.is_async = rows, True
        N.ntt(fname, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n, extension=True)
        torch.cuda.synchronize()
        pick = [1, 6]
        hx = np.ascontiguousarray(x[pick].cpu().numpy().view(np.uint32)).reshape(-1)
        hy = np.ascontiguousarray(y[pick].cpu().numpy().view(np.uint32)).reshape(-1)
        exp = rf.ntt(hx, n, 0, batch=len(pick), extension=True)
        assert np.array_equal(hy, exp), "extension_ntt forward 2^22 x 8: rows differ from the reference CPU backend"
        # every row: lane l of the extension transform is the base-field transform of lane l
        xl = x.permute(0, 2, 1).contiguous()  # [row][lane][element]
        yl = torch.empty_like(xl)
        cfgb = hip.NTTConfigU32.default()
        cfgb.batch_size, cfgb.is_async = rows * 4, True
        N.ntt(fname, xl.data_ptr(), N.FORWARD, cfgb, out=yl.data_ptr(), size=n)
        torch.cuda.synchronize()
        assert torch.equal(yl.permute(0, 2, 1), y)
        del xl, yl
        N.ntt(fname, y.data_ptr(), N.INVERSE, cfg, out=y.data_ptr(), size=n, extension=True)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        assert np.array_equal(rf.ntt(exp, n, 1, batch=len(pick), extension=True), hx)
        # coset + kNR, one row
        cfg1 = hip.NTTConfigU32.default()
        cfg1.batch_size, cfg1.is_async, cfg1.coset_gen, cfg1.ordering = 2, True, 7, N.kNR
        N.ntt(fname, x[2:4].data_ptr(), N.FORWARD, cfg1, out=y[2:4].data_ptr(), size=n, extension=True)
        torch.cuda.synchronize()
        r_in = np.ascontiguousarray(x[3].cpu().numpy().view(np.uint32)).reshape(-1)
        e = rf.ntt(r_in, n, 0, ordering=N.kNR, coset_gen=7, extension=True)
        assert np.array_equal(y[3].cpu().numpy().view(np.uint32).reshape(-1), e), "extension coset kNR forward"
    finally:
        N.release_domain(fname)
        rf.release_domain()


@pytest.mark.parametrize("fname", ["babybear", "koalabear"])
def test_ragged_lane_counts_whole_vs_oracle(hip, fname):
    """batch sizes that are not a power of two or not a multiple of the 32-lane slice (the Rust suite uses 100,
    wrappers/rust/icicle-core/src/ntt/tests.rs:276): partial slices, 1-lane tails, extension + columns; whole outputs
    against the reference, two- and three-pass sizes, every ordering."""
    from icicle_amd import ntt as N

    F = pyref.NTT_FIELDS[fname]
    rf = _domain(N, fname, 1 << 18)
    rng = np.random.default_rng(99)
    try:
        cases = [(16, 100, False, 0, 1, 0), (16, 33, False, 1, 1, 0), (17, 5, True, 0, 1, 1), (18, 12, False, 0, 9, 0), (14, 65, False, 1, 5, 1),
                 (12, 3, True, 2, 1, 0), (9, 100, False, 3, 3, 1), (8, 40, False, 0, 1, 0), (4, 100, True, 1, 1, 1), (17, 96, False, 1, 1, 0)]
        for logn, batch, ext, ordering, coset, direction in cases:
            n = 1 << logn
            lanes = 4 if ext else 1
            x = rng.integers(0, F.p, size=n * batch * lanes, dtype=np.uint32)
            cfg = hip.NTTConfigU32.default()
            cfg.batch_size, cfg.columns_batch, cfg.ordering, cfg.coset_gen = batch, True, ordering, coset
            got = N.ntt(fname, x, direction, cfg, extension=ext)
            exp = rf.ntt(x, n, direction, batch=batch, columns_batch=True, ordering=ordering, coset_gen=coset, extension=ext)
            assert np.array_equal(got, exp), (fname, logn, batch, ext, ordering, coset, direction)
    finally:
        N.release_domain(fname)
        rf.release_domain()
