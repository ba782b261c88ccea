#!/usr/bin/env python3
"""Timing matrix over the BASELINE configs' shapes on one GPU (device-resident inputs, hipEvents via
torch): prints one line per case. Used to fill profiles/rNN_perf_matrix.txt."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icicle_amd  # noqa: E402
from icicle_amd import msm as M, ntt as N, runtime  # noqa: E402
from icicle_amd._lib import MSMConfig, NTTConfigU32, NTTConfigU256, lib, check  # noqa: E402

runtime.set_device(0)
dev = torch.device("cuda", 0)
TOP = {"bn254": 0x30644E72, "bls12_381": 0x73EDA753, "bls12_377": 0x12AB655E, "grumpkin": 0x30644E72}  # top scalar word


def time_it(fn, reps=3):
    """ms per call. Warm-up runs for >= 40 ms of back-to-back calls: after an idle gap the first milliseconds of work run at
    ramping clocks (2^24 x 8 BabyBear: 0.89 ms after 3 warm-up calls, 0.71 ms after 100 -- profiles/r04_notes.md), which rounds
    1-3 booked against every sub-millisecond row of this matrix; then >= `reps` calls and >= 40 ms are timed."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one = None
    while time.perf_counter() - t0 < 0.04:
        fn()
        if one is None:
            torch.cuda.synchronize()
            one = max(1e-5, time.perf_counter() - t0)
    torch.cuda.synchronize()
    n = max(reps, min(200, int(0.04 / one) + 1))
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def msm_case(curve, logn, batch=1, pf=1, g2=False, c=0):
    n = 1 << logn
    L = M.LIMBS[curve] * (2 if g2 else 1)
    sym = f"{curve}_g2" if g2 else curve
    bases = torch.empty((n, 2 * L), dtype=torch.int32, device=dev)
    check(getattr(lib, f"{sym}_hip_generate_affine_points")(bases.data_ptr(), n, 1, True, None))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    sc = torch.randint(-(2 ** 31), 2 ** 31, (n * batch, 8), dtype=torch.int32, device=dev, generator=g)
    sc[:, 7] = torch.randint(0, TOP[curve], (n * batch,), dtype=torch.int32, device=dev, generator=g)
    res = torch.empty((batch, 3 * L), dtype=torch.int32, device=dev)
    cfg = MSMConfig.default()
    cfg.batch_size = batch
    cfg.is_async = True
    cfg.c = c
    ms = time_it(lambda: M.msm(curve, sc.data_ptr(), bases.data_ptr(), cfg, results=res.data_ptr(), msm_size=n, g2=g2))
    if c:
        print(f"msm {sym:12s} 2^{logn:<2d} c={c:<2d} {ms:9.3f} ms", flush=True)
        return
    print(f"msm {sym:12s} 2^{logn:<2d} batch {batch:<4d} {ms:9.3f} ms  {batch * n / ms / 1e6:8.2f} Gpoint/s", flush=True)


def ntt_case(field, logn, batch):
    n = 1 << logn
    N.init_domain(field, N.get_root_of_unity(field, n))
    p = {"babybear": 0x78000001, "koalabear": 0x7F000001}[field]
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    x = torch.randint(0, p, (batch, n), dtype=torch.int32, device=dev, generator=g)
    y = torch.empty_like(x)
    cfg = NTTConfigU32.default()
    cfg.batch_size = batch
    cfg.is_async = True
    ms = time_it(lambda: N.ntt(field, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n))
    gbs = 2 * batch * n * 4 / ms / 1e6
    print(f"ntt {field:10s} 2^{logn:<2d} batch {batch:<5d} {ms:9.3f} ms  {batch / ms * 1e3:10.0f} NTT/s  {gbs:7.0f} GB/s algorithmic", flush=True)
    N.release_domain(field)


def msm_distribution_case(dist, logn):
    """BN254 MSM under the reference's input distributions (tests/test_gpu_msm_distributions.py make_inputs): uniform,
    period-100 bases (projective.h:43-53), the Rust suite's skewed scalars (msm/tests.rs:256-276); phase split from the
    library's event timers"""
    import ctypes
    from tests.test_gpu_msm_distributions import make_inputs

    n = 1 << logn
    sc, bases = make_inputs(dist, logn, dev)
    res = torch.empty((1, 24), dtype=torch.int32, device=dev)
    cfg = MSMConfig.default()
    cfg.is_async = True
    run = lambda: M.msm("bn254", sc.data_ptr(), bases.data_ptr(), cfg, results=res.data_ptr(), msm_size=n)
    ms = time_it(run)
    lib.icicle_hip_enable_kernel_timing(True)
    tot, cnt = ctypes.c_double(), ctypes.c_int()
    for which in (0, 2, 3):
        lib.icicle_hip_kernel_timing(which, True, ctypes.byref(tot), ctypes.byref(cnt))
    run()
    torch.cuda.synchronize()
    ph = []
    for which, name in ((2, "digits+sort"), (0, "accumulate"), (3, "reduce+combine")):
        lib.icicle_hip_kernel_timing(which, True, ctypes.byref(tot), ctypes.byref(cnt))
        ph.append(f"{name} {tot.value:7.2f}")
    lib.icicle_hip_enable_kernel_timing(False)
    print(f"msm bn254 2^{logn:<2d} {dist:10s} {ms:9.3f} ms   [{', '.join(ph)} ms]", flush=True)


def ntt_layout_case(field, logn, batch, layout, ordering=0):
    """the interleaved layouts next to the row-major batch of the same byte count: layout = "rows" | "columns" (columns_batch,
    element j of transform b at j * batch + b) | "ext" (batch rows of quartic-extension elements = 4 * batch lane transforms)"""
    n = 1 << logn
    N.init_domain(field, N.get_root_of_unity(field, n))
    p = {"babybear": 0x78000001, "koalabear": 0x7F000001}[field]
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    lanes = 4 if layout == "ext" else 1
    x = torch.randint(0, p, (batch * n * lanes,), dtype=torch.int32, device=dev, generator=g)
    y = torch.empty_like(x)
    cfg = NTTConfigU32.default()
    cfg.batch_size = batch
    cfg.is_async = True
    cfg.columns_batch = layout == "columns"
    cfg.ordering = ordering
    row = []
    for d in (N.FORWARD, N.INVERSE):
        ms = time_it(lambda: N.ntt(field, x.data_ptr(), d, cfg, out=y.data_ptr(), size=n, extension=layout == "ext"))
        row.append(f"{'fwd' if d == N.FORWARD else 'inv'} {ms:8.3f} ms {2 * x.numel() * 4 / ms / 1e6:6.0f} GB/s")
    print(f"ntt {field:10s} 2^{logn:<2d} x {batch:<5d} {layout:8s} ordering {ordering}: " + ", ".join(row), flush=True)
    N.release_domain(field)


def ntt_polymul_case(field, logn, batch):
    """the polynomial-product pipeline the orderings exist for: two kNR forward transforms, a point-wise product
    in bit-reversed order, one kRN inverse transform -- no reordering anywhere, everything device resident"""
    from icicle_amd import vecops as V
    from icicle_amd._lib import VecOpsConfig

    n = 1 << logn
    N.init_domain(field, N.get_root_of_unity(field, n))
    p = {"babybear": 0x78000001, "koalabear": 0x7F000001}[field]
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    a = torch.randint(0, p, (batch, n), dtype=torch.int32, device=dev, generator=g)
    b = torch.randint(0, p, (batch, n), dtype=torch.int32, device=dev, generator=g)
    fa, fb, c = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    cf = NTTConfigU32.default()
    cf.batch_size, cf.is_async, cf.ordering = batch, True, N.kNR
    ci = NTTConfigU32.default()
    ci.batch_size, ci.is_async, ci.ordering = batch, True, N.kRN
    vc = VecOpsConfig.default()
    vc.is_async = True

    def run():
        N.ntt(field, a.data_ptr(), N.FORWARD, cf, out=fa.data_ptr(), size=n)
        N.ntt(field, b.data_ptr(), N.FORWARD, cf, out=fb.data_ptr(), size=n)
        V.vector_mul(field, fa.data_ptr(), fb.data_ptr(), vc, out=fa.data_ptr(), size=batch * n)
        N.ntt(field, fa.data_ptr(), N.INVERSE, ci, out=c.data_ptr(), size=n)

    ms = time_it(run)
    one = time_it(lambda: N.ntt(field, a.data_ptr(), N.FORWARD, NTT_NN(batch), out=fa.data_ptr(), size=n))
    print(f"polymul {field:10s} 2^{logn:<2d} x {batch:<4d} kNR + kNR + vector_mul + kRN(inverse): {ms:8.3f} ms   (one kNN forward: {one:7.3f} ms -> pipeline / 3 transforms = {ms / 3 / one:5.2f} x kNN)", flush=True)
    N.release_domain(field)


def NTT_NN(batch):
    cfg = NTTConfigU32.default()
    cfg.batch_size, cfg.is_async = batch, True
    return cfg


def ntt_scalar_case(field, logn, batch):
    """NTT over the curve's 256-bit scalar field; inputs are any words < 2^253 (valid canonical elements)"""
    n = 1 << logn
    N.init_domain(field, N.get_root_of_unity(field, n))
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randint(-(2 ** 31), 2 ** 31, (batch, n, 8), dtype=torch.int32, device=dev, generator=g)
    x[:, :, 7] &= 0x0FFFFFFF
    y = torch.empty_like(x)
    cfg = NTTConfigU256.default()
    cfg.batch_size = batch
    cfg.is_async = True
    ms = time_it(lambda: N.ntt(field, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n))
    gbs = 2 * batch * n * 32 / ms / 1e6
    print(f"ntt {field + '_fr':12s} 2^{logn:<2d} batch {batch:<5d} {ms:9.3f} ms  {batch * n / ms / 1e6:8.2f} Gelem/s  {gbs:7.0f} GB/s algorithmic", flush=True)
    N.release_domain(field)


def msm_precompute_case(curve, logn, pf, batch=1, c=0):
    """the reference's precompute sweep (docs/docs/api/cpp/msm.md:186-201, wrappers/rust/icicle-core/src/msm/mod.rs:386-470):
    msm_precompute_bases once, then timed msm() calls with precompute_factor = pf (shared bases over the batch)"""
    n = 1 << logn
    L = M.LIMBS[curve]
    bases = torch.empty((n, 2 * L), dtype=torch.int32, device=dev)
    check(getattr(lib, f"{curve}_hip_generate_affine_points")(bases.data_ptr(), n, 1, True, None))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    sc = torch.randint(-(2 ** 31), 2 ** 31, (n * batch, 8), dtype=torch.int32, device=dev, generator=g)
    sc[:, 7] = torch.randint(0, TOP[curve], (n * batch,), dtype=torch.int32, device=dev, generator=g)
    res = torch.empty((batch, 3 * L), dtype=torch.int32, device=dev)
    cfg = MSMConfig.default()
    cfg.batch_size = batch
    cfg.is_async = True
    cfg.precompute_factor = pf
    cfg.c = c
    table = bases
    t_pre = 0.0
    if pf > 1:
        table = torch.empty((n * pf, 2 * L), dtype=torch.int32, device=dev)
        t_pre = 1e30
        for _ in range(2):  # (the first call also pays for the code object load and the workspace lease)
            t0 = time.perf_counter()
            M.precompute_bases(curve, bases.data_ptr(), cfg, output=table.data_ptr(), nof_bases=n)
            torch.cuda.synchronize()
            t_pre = min(t_pre, (time.perf_counter() - t0) * 1e3)
    ms = time_it(lambda: M.msm(curve, sc.data_ptr(), table.data_ptr(), cfg, results=res.data_ptr(), msm_size=n))
    import ctypes
    pc, pw = ctypes.c_int(), ctypes.c_int()
    check(lib.icicle_hip_msm_plan(n, 254, ctypes.byref(cfg), ctypes.byref(pc), ctypes.byref(pw)))
    print(f"msm {curve:10s} 2^{logn:<2d} batch {batch:<4d} precompute_factor {pf} c = {pc.value:<2d}{' (forced)' if c else '         '} {ms:9.3f} ms  (precompute_bases {t_pre:8.2f} ms)", flush=True)


def ecntt_case(curve, logn, batch=1):
    n = 1 << logn
    L = M.LIMBS[curve]
    N.init_domain(curve, N.get_root_of_unity(curve, n))
    aff = torch.empty((n * batch, 2 * L), dtype=torch.int32, device=dev)
    check(getattr(lib, f"{curve}_hip_generate_affine_points")(aff.data_ptr(), n * batch, 7, True, None))
    one = torch.zeros((n * batch, L), dtype=torch.int32, device=dev)
    one[:, 0] = 1
    x = torch.cat([aff, one], dim=1).contiguous()  # projective_t (x, y, 1)
    y = torch.empty_like(x)
    cfg = NTTConfigU256.default()
    cfg.batch_size = batch
    cfg.is_async = True
    ms = time_it(lambda: N.ecntt(curve, x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=n), reps=2)
    print(f"ecntt {curve:10s} 2^{logn:<2d} batch {batch:<3d} {ms:10.2f} ms  ({n // 2 * logn * batch} butterflies, {ms * 1e3 / max(1, logn):8.1f} us per stage)", flush=True)
    N.release_domain(curve)



def criterion_sweep():
    """The shape of the reference's own criterion benches (wrappers/rust/icicle-core/src/msm/mod.rs:386-470: sizes
    2^13..2^25, precompute_factor {1, 4, 8}, batch {1, 16, 128} with batch * size <= 2^25, HOST scalars, precomputed bases
    on the device, async on a stream; ntt/mod.rs:504-572: sizes from 2^13, batches from 2^7 with batch * size <= 2^25,
    host in / out, both directions, all six orderings), thinned to every fourth size."""
    import numpy as np

    for logn in (13, 17, 21, 25):
        n = 1 << logn
        bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
        check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, 1, True, None))
        for pf in (1, 4, 8):
            cfg = MSMConfig.default()
            cfg.precompute_factor = pf
            cfg.is_async = True
            table = bases
            if pf > 1:
                table = torch.empty((n * pf, 16), dtype=torch.int32, device=dev)
                M.precompute_bases("bn254", bases.data_ptr(), cfg, output=table.data_ptr(), nof_bases=n)
            for batch in (1, 16, 128):
                if batch * n > (1 << 25):
                    continue
                rng = np.random.default_rng(logn * 10 + batch)
                hs = rng.integers(0, 1 << 32, size=(batch * n, 8), dtype=np.uint64).astype(np.uint32)
                hs[:, 7] &= 0x0FFFFFFF
                res = torch.empty((batch, 24), dtype=torch.int32, device=dev)
                cfg.batch_size = batch
                ms = time_it(lambda: M.msm("bn254", hs, table.data_ptr(), cfg, results=res.data_ptr(), msm_size=n))
                print(f"criterion msm bn254 {n} x {batch} with precomp = {pf}: {ms:9.3f} ms  (host scalars, device bases)", flush=True)
            del table
        del bases
    N.init_domain("babybear", N.get_root_of_unity("babybear", 1 << 21))
    names = ["kNN", "kNR", "kRN", "kRR", "kNM", "kMN"]
    for logn, batch in ((13, 128), (13, 4096), (17, 128), (21, 16)):
        n = 1 << logn
        rng = np.random.default_rng(logn)
        hx = rng.integers(0, 0x78000001, size=(batch, n), dtype=np.uint32)
        hy = np.empty_like(hx)
        row = []
        for d in (N.FORWARD, N.INVERSE):
            for o in range(6):
                cfg = NTTConfigU32.default()
                cfg.batch_size = batch
                cfg.ordering = o
                ms = time_it(lambda: N.ntt("babybear", hx, d, cfg, out=hy))
                row.append(f"{names[o]} {'fwd' if d == N.FORWARD else 'inv'} {ms:7.3f}")
        print(f"criterion ntt babybear {n} x {batch} (host in / out, ms): " + ", ".join(row), flush=True)
    N.release_domain("babybear")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "csweep":
        sweep = ((10, (7, 8, 9, 10, 11)), (12, (8, 9, 10, 11, 12)), (14, (9, 10, 11, 12, 13, 14)), (16, (10, 11, 12, 13, 14, 15)), (18, (12, 13, 14, 15, 16, 17)), (20, (14, 15, 16, 17, 18)),
                 (21, (15, 16, 17, 18, 19)), (22, (15, 16, 17, 18, 19)), (23, (16, 17, 18, 19, 20)), (24, (17, 18, 19, 20, 21)), (25, (18, 19, 20, 21, 22)), (26, (19, 20, 21, 22)))
        if len(sys.argv) > 2:
            sweep = tuple(x for x in sweep if x[0] == int(sys.argv[2]))
        for logn, cs in sweep:
            msm_case("bn254", logn)
            for c in cs:
                msm_case("bn254", logn, c=c)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "host":
        # PCIe-inclusive: operands handed over as host (pageable numpy) buffers, result back on the host
        import numpy as np

        for logn in (20, 24, 26):
            n = 1 << logn
            bases_d = torch.empty((n, 16), dtype=torch.int32, device=dev)
            check(lib.bn254_hip_generate_affine_points(bases_d.data_ptr(), n, 1, True, None))
            hb = bases_d.cpu().numpy().view(np.uint32)
            rng = np.random.default_rng(0)
            hs = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
            hs[:, 7] &= 0x0FFFFFFF
            # host scalars (the wrappers' HostSlice), bases resident on the device: chunks of scalars are uploaded behind the
            # previous chunk's MSM (msm_multi.hpp); pageable and pinned (torch pin_memory) host memory
            out = np.zeros((1, 24), dtype=np.uint32)
            for label, src in (("pageable", hs), ("pinned", torch.from_numpy(hs.view(np.int32)).pin_memory().numpy().view(np.uint32))):
                M.msm("bn254", src, bases_d.data_ptr(), MSMConfig.default(), results=out, msm_size=n)
                t0 = time.perf_counter()
                M.msm("bn254", src, bases_d.data_ptr(), MSMConfig.default(), results=out, msm_size=n)
                ms = (time.perf_counter() - t0) * 1e3
                print(f"msm bn254 2^{logn} host scalars ({label}), device-resident bases: {ms:9.2f} ms  ({hs.nbytes / 1e9:.2f} GB of scalars over PCIe)", flush=True)
            del bases_d
            M.msm("bn254", hs, hb)
            t0 = time.perf_counter()
            M.msm("bn254", hs, hb)
            ms = (time.perf_counter() - t0) * 1e3
            gb = (hs.nbytes + hb.nbytes) / 1e9
            print(f"msm bn254 2^{logn} host-resident operands: {ms:9.2f} ms  ({gb:.2f} GB over PCIe, {gb / ms * 1e3:.1f} GB/s effective)", flush=True)
            del hs, hb
        N.init_domain("babybear", N.get_root_of_unity("babybear", 1 << 24))
        for batch in (8, 64):
            rng = np.random.default_rng(1)
            hx = rng.integers(0, 0x78000001, size=(batch, 1 << 24), dtype=np.uint32)
            cfg = NTTConfigU32.default()
            cfg.batch_size = batch
            for label, src in (("pageable", hx), ("pinned", torch.from_numpy(hx.view(np.int32)).pin_memory().numpy().view(np.uint32))):
                # the output buffer exists and has been touched before the timed call (a fresh np.zeros costs a page fault per
                # 4 KiB on its first write: 0.3 s for 4 GiB, which is the caller's allocation, not the transform)
                dst = np.ones_like(src) if label == "pageable" else torch.empty(src.shape, dtype=torch.int32).pin_memory().numpy().view(np.uint32)
                N.ntt("babybear", src, N.FORWARD, cfg, out=dst)
                t0 = time.perf_counter()
                N.ntt("babybear", src, N.FORWARD, cfg, out=dst)
                ms = (time.perf_counter() - t0) * 1e3
                print(f"ntt babybear 2^24 x {batch} host-resident in/out ({label}): {ms:9.2f} ms  ({2 * hx.nbytes / ms / 1e6:.1f} GB/s effective both ways)", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ntt":
        for logn, batch in ((12, 4096), (16, 1024), (20, 256), (22, 128), (24, 64), (24, 8), (25, 32), (26, 16), (27, 8), (27, 4)):
            ntt_case("babybear", logn, batch)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pfsweep":  # window size against precompute_factor: the fit behind msm_plan.h
        for logn, cs in ((16, (11, 12, 13, 15, 16)), (18, (13, 15, 16, 17)), (20, (15, 16, 17, 19, 20)), (22, (16, 17, 19, 20)), (24, (17, 19, 20))):
            for pf in (4, 8):
                msm_precompute_case("bn254", logn, pf)
                for c in cs:
                    msm_precompute_case("bn254", logn, pf, c=c)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "distributions":
        for logn in (22, 26):
            for dist in ("uniform", "period100", "skewed"):
                msm_distribution_case(dist, logn)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "layouts":  # interleaved layouts vs the row-major batch of the same bytes
        for field, logn, b in (("koalabear", 22, 64), ("babybear", 22, 32), ("babybear", 16, 1024), ("babybear", 24, 16), ("babybear", 20, 100)):
            ntt_layout_case(field, logn, b, "rows")
            ntt_layout_case(field, logn, b, "columns")
            if b % 4 == 0:
                ntt_layout_case(field, logn, b // 4, "ext")
        ntt_layout_case("babybear", 22, 8, "ext")
        for layout in ("rows", "columns"):  # bit-reversed output / input
            ntt_layout_case("babybear", 22, 64, layout, ordering=1)
            ntt_layout_case("babybear", 22, 64, layout, ordering=2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "polymul":
        for field, logn, b in (("babybear", 24, 64), ("babybear", 22, 64), ("koalabear", 22, 64), ("babybear", 16, 1024)):
            ntt_polymul_case(field, logn, b)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "criterion":
        criterion_sweep()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "precompute":
        for logn, batch in ((16, 1), (16, 16), (20, 1), (20, 16), (22, 1), (24, 1)):
            for pf in (1, 4, 8):
                msm_precompute_case("bn254", logn, pf, batch)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ecntt":
        for logn in (8, 10, 11, 12, 13, 14, 16):
            ecntt_case("bn254", logn)
        ecntt_case("bls12_381", 10)
        ecntt_case("bn254", 10, batch=8)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "groups":  # A/B of the pipelined schedule: run once per ICICLE_HIP_MSM_GROUPS value
        print("ICICLE_HIP_MSM_GROUPS =", os.environ.get("ICICLE_HIP_MSM_GROUPS", "(default)"), " SORT_CUS =", os.environ.get("ICICLE_HIP_MSM_SORT_CUS", "-"), flush=True)
        for logn in (12, 16, 20, 22, 24, 26):
            msm_case("bn254", logn)
        msm_case("bls12_381", 24)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "midsize":
        for logn in (16, 18, 20, 21, 22, 23, 24):
            msm_case("bn254", logn)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "curves":  # the further curves / fields next to the two of the BASELINE configs
        for curve in ("bn254", "grumpkin", "bls12_381", "bls12_377"):
            for logn in (16, 20, 22):
                msm_case(curve, logn)
        msm_case("bls12_381", 18, g2=True)
        msm_case("bls12_377", 18, g2=True)
        for field in ("bn254", "bls12_377", "stark252"):
            ntt_scalar_case(field, 20, 4)
        ecntt_case("bls12_377", 10)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g2":
        for logn in (12, 16, 20, 22, 24):
            msm_case("bn254", logn, g2=True)
        for logn in (16, 20, 22):
            msm_case("bls12_381", logn, g2=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "scalar-ntt":
        for logn, batch in ((12, 256), (16, 16), (20, 1), (22, 1), (24, 1)):
            ntt_scalar_case("bn254", logn, batch)
        ntt_scalar_case("bls12_381", 22, 1)
        sys.exit(0)
    for logn in (12, 16, 20, 22, 24, 26):
        msm_case("bn254", logn)
    msm_case("bn254", 12, batch=1024)
    for logn in (20, 24, 25):
        msm_case("bls12_381", logn)
    msm_case("bls12_381", 12, batch=1024)  # (the reference publishes 136.7 ms for this shape on an RTX 3090 Ti)
    for logn, batch in ((12, 4096), (16, 1024), (20, 256), (24, 64), (27, 4)):
        ntt_case("babybear", logn, batch)
    for logn, batch in ((22, 128), (22, 1024), (24, 64)):
        ntt_case("koalabear", logn, batch)
