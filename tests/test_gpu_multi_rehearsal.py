"""GPU: the multi-device code behind the C ABI with P > 1 device SLOTS on the one GPU of the test box (VERDICT r02
items 2, 3).

Real RCCL refuses one GPU twice in a communicator, so until now every "G = 8" test ran one worker: no host thread per
device, no gate crossing, no peer copy, and the grouped ncclSend / ncclRecv of the bucket exchange had zero executions.
Here icicle_hip_test_set_virtual_devices(K) maps K device slots onto GPU 0 and the loopback stand-in (a library of its own since
round 4: tests/loopback/nccl_loopback.cpp -> tests/_build/libnccl_loopback.so, handed to the product with
icicle_hip_set_collectives_library) takes RCCL's place with the same symbols, call sequence and stream-ordering contract:
  * msm, hip_num_devices = G on P in {2, 8} slots: one host thread + stream + communicator rank per slot, operands of the
    slots other than 0 staged by (peer-style) copies through the two-slot ring, E1 all-gather + k_proj_sum and E2 grouped
    send / recv of bucket slices + k_bucket_add, against the reference CPU backend and against the single-call MSM;
  * a worker that fails before each gate (set-up, bucket exchange, result gather): the call returns an error, nobody
    hangs, and the next call works;
  * the bucket exchange with a tiny window (c <= 6: fewer buckets than one reduction chunk -- the out-of-window slice
    ADVICE r02 found);
  * "hip_bases_resident": the second call moves 0 bytes of bases;
  * batched NTT row shards on P slots (31-bit fields and the 256-bit scalar field), the three-stream host pipeline;
  * chunked host-scalar MSM (the single-GPU HostSlice path) against the device-resident call.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import pyref, ref
from tests.util import cached_points, points_to_array, rand_scalars, to_words

# a dead-locked collective must cost two minutes, not the whole run (the main thread then sits in a C call: only the
# watchdog-thread method can end it)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180, method="thread")]


@pytest.fixture()
def slots(hip):
    """yields a function that switches to K virtual slots + loopback collectives; always restored afterwards"""
    from icicle_amd._lib import lib, check

    loopback = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libnccl_loopback.so")
    if not os.path.exists(loopback):
        pytest.skip("tests/_build/libnccl_loopback.so not built (__graft_entry__.build())")

    def use(k):
        check(lib.icicle_hip_test_set_virtual_devices(k))
        check(lib.icicle_hip_set_collectives_library(loopback.encode() if k > 0 else None))

    yield use
    lib.icicle_hip_test_inject_failure(-1, 0)
    lib.icicle_hip_test_set_virtual_devices(0)
    lib.icicle_hip_set_collectives_library(None)
    lib.icicle_hip_msm_release_resident_bases(None)


def _ext(**kv):
    from icicle_amd._lib import lib

    e = lib.create_config_extension()
    for k, v in kv.items():
        if isinstance(v, bool):
            lib.config_extension_set_bool(e, k.encode(), v)
        else:
            lib.config_extension_set_int(e, k.encode(), v)
    return e


def _inputs(cname, n, seed):
    C = pyref.CURVES[cname]
    rng = np.random.default_rng(seed)
    pts = list(cached_points(C, n))
    pts[1] = pyref.INF
    bases = points_to_array(C, pts)
    sc = to_words(rand_scalars(rng, n, C.r), 8)
    return C, rng, bases, sc


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
@pytest.mark.parametrize("P,G", [(2, 2), (2, 5), (8, 8), (8, 11)])
@pytest.mark.parametrize("mode", ["partial_sums", "bucket_exchange"])
def test_msm_on_p_device_slots(hip, slots, cname, P, G, mode):
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, multi_stats
    from icicle_amd.runtime import DeviceVec

    if cname == "bls12_381" and (P, G) not in ((2, 5), (8, 8)):
        pytest.skip("second curve: two shapes are enough")
    C, rng, bases, sc = _inputs(cname, 6007, 300 + P + G)
    n = len(sc)
    refc = ref.RefCurve(cname)
    exp = refc.to_affine(refc.msm(sc, bases))
    single = M.msm(cname, sc, bases)
    assert np.array_equal(refc.to_affine(single), exp)
    slots(P)
    ext = _ext(hip_num_devices=G, hip_msm_exchange_buckets=(mode == "bucket_exchange"))
    try:
        multi_stats(reset=True)
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        got = M.msm(cname, sc, bases, cfg)  # host operands: every slot uploads its own shards
        assert np.array_equal(refc.to_affine(got), exp), (cname, P, G, mode, "host")
        assert refc.is_on_curve(got[0])
        st = multi_stats()
        assert st["threaded_calls"] == 1 and st["staged_scalar_bytes"] == n * 32
        if mode == "bucket_exchange":
            assert st["exchanged_bucket_bytes"] > 0, "the grouped send / recv of the bucket exchange did not run"
            # at most ONE message per ordered peer pair (rounds 2-3: one per window as well); a slot whose slice is empty -- fewer
            # reduction chunks than slots at this size -- receives none
            assert 1 <= st["exchange_messages"] <= P * (P - 1), st
        # device-resident operands on the calling device, batch of 2 with shared bases: slots > 0 copy from slot 0's device
        sc2 = np.vstack([sc, to_words(rand_scalars(rng, n, C.r), 8)])
        d_sc, d_b = DeviceVec.from_host(sc2), DeviceVec.from_host(bases)
        cfg2 = hip.MSMConfig.default()
        cfg2.ext = ext
        cfg2.batch_size = 2
        cfg2.are_points_shared_in_batch = True
        got2 = M.msm(cname, d_sc, d_b, cfg2, msm_size=n)
        exp2 = refc.to_affine(refc.msm(sc2, bases, batch=2, shared=True))
        assert np.array_equal(refc.to_affine(got2), exp2), (cname, P, G, mode, "device batch")
    finally:
        lib.destroy_config_extension(ext)


@pytest.mark.parametrize("mode", ["partial_sums", "bucket_exchange"])
@pytest.mark.parametrize("stage", [1, 2, 3])
@pytest.mark.parametrize("victim", [0, 2])
def test_worker_failure_before_each_gate(hip, slots, mode, stage, victim):
    """slot `victim` fails at set-up (1), right before the bucket-exchange gate (2) or right before the result-gather
    gate (3): the call must come back with an error -- the peers skip the collective instead of waiting in it -- and the
    same call without the fault must then succeed."""
    from icicle_amd import msm as M
    from icicle_amd._lib import IcicleError, lib, check

    if stage == 2 and mode != "bucket_exchange":
        pytest.skip("stage 2 is the bucket-exchange gate")
    C, rng, bases, sc = _inputs("bn254", 5003, 77)
    refc = ref.RefCurve("bn254")
    exp = refc.to_affine(refc.msm(sc, bases))
    slots(4)
    ext = _ext(hip_num_devices=4, hip_msm_exchange_buckets=(mode == "bucket_exchange"))
    try:
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        check(lib.icicle_hip_test_inject_failure(victim, stage))
        with pytest.raises(IcicleError):
            M.msm("bn254", sc, bases, cfg)
        got = M.msm("bn254", sc, bases, cfg)  # one-shot fault: gone
        assert np.array_equal(refc.to_affine(got), exp)
    finally:
        lib.destroy_config_extension(ext)


@pytest.mark.parametrize("c", [3, 5, 6, 8])
@pytest.mark.parametrize("P", [1, 2, 3])
def test_bucket_exchange_with_tiny_windows(hip, slots, c, P):
    """with c <= 6 a window has fewer buckets than one reduction chunk (64); the slice sizes of the
    exchange must be clamped to the window."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib

    C, rng, bases, sc = _inputs("bn254", 700, 500 + c)
    refc = ref.RefCurve("bn254")
    exp = refc.to_affine(refc.msm(sc, bases))
    slots(P if P > 1 else 0)
    ext = _ext(hip_num_devices=3, hip_msm_exchange_buckets=True)
    try:
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        cfg.c = c
        got = M.msm("bn254", sc, bases, cfg)
        assert np.array_equal(refc.to_affine(got), exp), (c, P)
    finally:
        lib.destroy_config_extension(ext)


@pytest.mark.parametrize("where", ["host", "device"])
def test_resident_bases_second_call_moves_no_bases(hip, slots, where):
    """bases stay resident per GPU, scalars are streamed. With "hip_bases_resident" the first call places
    the base shards on their devices; the second call with the same base pointer stages 0 bytes of bases."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check, multi_stats
    from icicle_amd.runtime import DeviceVec

    C, rng, bases, sc = _inputs("bn254", 9001, 901)
    n = len(sc)
    refc = ref.RefCurve("bn254")
    slots(4)
    ext = _ext(hip_num_devices=8, hip_bases_resident=True)
    try:
        b = DeviceVec.from_host(bases) if where == "device" else bases
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        multi_stats(reset=True)
        got = M.msm("bn254", sc, b, cfg, msm_size=n)
        first = multi_stats(reset=True)
        assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases)))
        # slot 0 gathers device-resident bases in place; every other shard (host bases: all shards) is placed once
        expect_first = n * 64 if where == "host" else None
        if expect_first is not None:
            assert first["staged_base_bytes"] == expect_first
        assert first["staged_base_bytes"] > 0
        sc2 = to_words(rand_scalars(rng, n, C.r), 8)
        got2 = M.msm("bn254", sc2, b, cfg, msm_size=n)
        second = multi_stats(reset=True)
        assert second["staged_base_bytes"] == 0, second
        assert second["resident_base_hits"] > 0 and second["staged_scalar_bytes"] == n * 32
        assert np.array_equal(refc.to_affine(got2), refc.to_affine(refc.msm(sc2, bases)))
        check(lib.icicle_hip_msm_release_resident_bases(b.ptr if where == "device" else bases.ctypes.data))
        M.msm("bn254", sc2, b, cfg, msm_size=n)
        assert multi_stats(reset=True)["staged_base_bytes"] == first["staged_base_bytes"]  # released: placed again
    finally:
        lib.destroy_config_extension(ext)


def test_host_scalars_are_pipelined_in_chunks(hip):
    """the wrappers' default HostSlice scalars on one GPU: n >= 2^22 is cut into chunks whose uploads hide behind the
    previous chunk's MSM (msm_multi.hpp); same group element as the device-resident single call and the reference"""
    import torch
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, check, multi_stats

    n = (1 << 22) + 12345
    dev = torch.device("cuda", 0)
    bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
    check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, 4242, True, None))
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    sc = torch.randint(-(2 ** 31), 2 ** 31, (n, 8), dtype=torch.int32, device=dev, generator=g)
    sc[:, 7] = torch.randint(0, 0x30644E72, (n,), dtype=torch.int32, device=dev, generator=g)
    torch.cuda.synchronize()
    hs = np.ascontiguousarray(sc.cpu().numpy().view(np.uint32))
    refc = ref.RefCurve("bn254")
    dev_res = np.zeros((1, 24), dtype=np.uint32)
    M.msm("bn254", sc.data_ptr(), bases.data_ptr(), hip.MSMConfig.default(), results=dev_res, msm_size=n)
    multi_stats(reset=True)
    host_res = M.msm("bn254", hs, bases.data_ptr(), hip.MSMConfig.default(), msm_size=n)
    st = multi_stats()
    assert st["staged_scalar_bytes"] == n * 32 and st["threaded_calls"] == 0
    assert np.array_equal(refc.to_affine(host_res), refc.to_affine(dev_res))
    hb = np.ascontiguousarray(bases.cpu().numpy().view(np.uint32))
    assert np.array_equal(refc.to_affine(host_res), refc.to_affine(refc.msm(hs, hb)))


@pytest.mark.parametrize("fname", ["koalabear", "babybear"])
@pytest.mark.parametrize("P,G", [(2, 2), (8, 8), (3, 7)])
def test_ntt_row_shards_on_p_device_slots(hip, slots, fname, P, G):
    """batched NTT, hip_num_devices = G on P slots: one host thread + stream per slot, rows staged through the
    three-buffer ring (slots > 0 treat the caller's buffers as remote), memcmp against the reference CPU backend"""
    from icicle_amd import ntt as N
    from icicle_amd._lib import lib
    from icicle_amd.runtime import DeviceVec

    F = pyref.NTT_FIELDS[fname]
    logn, batch = 11, 13
    n = 1 << logn
    rng = np.random.default_rng(P * 10 + G)
    x = rng.integers(0, F.p, size=batch * n, dtype=np.uint32)
    rf = ref.RefNttField(fname)
    root = N.get_root_of_unity(fname, n)
    rf.init_domain(root)
    N.init_domain(fname, root)
    ext = _ext(hip_num_devices=G)
    try:
        exp = rf.ntt(x, n, 0, batch=batch)
        slots(P)
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size = batch
        cfg.ext = ext
        y = N.ntt(fname, x, N.FORWARD, cfg)  # host in / out
        assert np.array_equal(y, exp), (fname, P, G, "host")
        dx, dy = DeviceVec.from_host(x), DeviceVec(x.nbytes)
        N.ntt(fname, dx, N.FORWARD, cfg, out=dy, size=n)  # device in / out on the calling device
        assert np.array_equal(dy.to_host(shape=x.shape), exp), (fname, P, G, "device")
        cfg.ordering = N.kNR
        cfg.coset_gen = 7
        z = N.ntt(fname, x, N.FORWARD, cfg)
        assert np.array_equal(z, rf.ntt(x, n, 0, batch=batch, ordering=N.kNR, coset_gen=7))
        back = N.ntt(fname, y, N.INVERSE, _plain(hip, batch, ext))
        assert np.array_equal(back, x)
    finally:
        lib.destroy_config_extension(ext)
        N.release_domain(fname)
        rf.release_domain()


def _plain(hip, batch, ext):
    cfg = hip.NTTConfigU32.default()
    cfg.batch_size = batch
    cfg.ext = ext
    return cfg


def test_scalar_field_ntt_row_shards(hip, slots):
    """"hip_num_devices" is honoured by the 256-bit scalar-field NTT as well"""
    from icicle_amd import ntt as N
    from icicle_amd._lib import lib, multi_stats
    from tests.test_gpu_ntt_scalar import rand_elems

    fname = "bn254"
    F = pyref.NTT_FIELDS[fname]
    logn, batch = 9, 10
    n = 1 << logn
    rng = np.random.default_rng(5)
    x = rand_elems(rng, F.p, batch * n)
    rf = ref.RefScalarNttField(fname)
    root = N.get_root_of_unity(fname, n)
    rf.init_domain(root)
    N.init_domain(fname, root)
    ext = _ext(hip_num_devices=4)
    try:
        slots(4)
        cfg = hip.NTTConfigU256.default()
        cfg.batch_size = batch
        cfg.ext = ext
        multi_stats(reset=True)
        y = N.ntt(fname, x, N.FORWARD, cfg)
        assert multi_stats()["threaded_calls"] == 1
        assert np.array_equal(y, rf.ntt(x, n, 0, batch=batch))
        assert np.array_equal(N.ntt(fname, y, N.INVERSE, cfg), x)
    finally:
        lib.destroy_config_extension(ext)
        N.release_domain(fname)
        rf.release_domain()


def test_host_resident_ntt_batch_is_pipelined(hip):
    """host in / out, 128 MiB batch on one GPU: row groups through the upload / compute / download streams
    (the shape of the reference's examples/c++/best-practice-ntt), memcmp against the reference on sampled rows"""
    from icicle_amd import ntt as N
    from icicle_amd._lib import multi_stats

    fname = "babybear"
    F = pyref.NTT_FIELDS[fname]
    logn, batch = 20, 32
    n = 1 << logn
    rng = np.random.default_rng(8)
    x = rng.integers(0, F.p, size=batch * n, dtype=np.uint32)
    rf = ref.RefNttField(fname)
    root = N.get_root_of_unity(fname, n)
    rf.init_domain(root)
    N.init_domain(fname, root)
    try:
        cfg = hip.NTTConfigU32.default()
        cfg.batch_size = batch
        multi_stats(reset=True)
        y = N.ntt(fname, x, N.FORWARD, cfg)
        st = multi_stats()
        assert st["staged_scalar_bytes"] == x.nbytes and st["threaded_calls"] == 0, st  # went through the row-group pipeline
        for r in (0, 7, 8, 19, 31):
            assert np.array_equal(y[r * n:(r + 1) * n], rf.ntt(np.ascontiguousarray(x[r * n:(r + 1) * n]), n, 0)), r
        z = x.copy()
        N.ntt(fname, z, N.FORWARD, cfg, out=z)  # in place on the host buffer
        assert np.array_equal(z, y)
        assert np.array_equal(N.ntt(fname, y, N.INVERSE, cfg), x)
    finally:
        N.release_domain(fname)
        rf.release_domain()


@pytest.mark.parametrize("fname", ["babybear", "koalabear"])
@pytest.mark.parametrize("P,logn", [(2, 6), (2, 11), (4, 9), (4, 14), (8, 8), (8, 16), (8, 20)])
def test_single_ntt_split_over_device_slots(hip, slots, fname, P, logn):
    """hip_num_devices = P with FEWER transforms than devices: every transform is cut over the P slots (four-step, three
    all-to-all exchanges with grouped send / recv, icicle_amd/csrc/ntt_split.hpp) -- the in-library form of the north
    star's "all-to-all of NTT chunks". memcmp against the reference CPU backend, forward and inverse, batch 1 and 3."""
    from icicle_amd import ntt as N
    from icicle_amd._lib import lib, multi_stats
    from icicle_amd.runtime import DeviceVec

    if fname == "koalabear" and (P, logn) not in ((2, 11), (8, 16)):
        pytest.skip("second field: two shapes are enough")
    F = pyref.NTT_FIELDS[fname]
    n = 1 << logn
    rng = np.random.default_rng(P * 100 + logn)
    rf = ref.RefNttField(fname)
    root = N.get_root_of_unity(fname, n)
    rf.init_domain(root)
    N.init_domain(fname, root)
    ext = _ext(hip_num_devices=P)
    try:
        slots(P)
        for batch in ((1, 3) if logn <= 16 else (1,)):
            if batch >= P:
                continue
            x = rng.integers(0, F.p, size=batch * n, dtype=np.uint32)
            exp = rf.ntt(x, n, 0, batch=batch)
            cfg = hip.NTTConfigU32.default()
            cfg.batch_size = batch
            cfg.ext = ext
            multi_stats(reset=True)
            y = N.ntt(fname, x, N.FORWARD, cfg)  # host in / out
            st = multi_stats()
            assert st["threaded_calls"] == 1 and st["exchanged_bucket_bytes"] == 3 * batch * P * (P - 1) * (n // (P * P)) * 4, st  # the split path ran: 3 exchanges
            assert np.array_equal(y, exp), (fname, P, logn, batch, "forward")
            back = N.ntt(fname, y, N.INVERSE, cfg)
            assert np.array_equal(back, x), (fname, P, logn, batch, "inverse")
            dx, dy = DeviceVec.from_host(x), DeviceVec(x.nbytes)
            N.ntt(fname, dx, N.FORWARD, cfg, out=dy, size=n)  # operands on the calling device
            assert np.array_equal(dy.to_host(shape=x.shape), exp)
        # a configuration the split does not cover (bit-reversed output) falls back to the row-shard path: same bytes as the reference
        x = rng.integers(0, F.p, size=n, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.ext = ext
        cfg.ordering = N.kNR
        assert np.array_equal(N.ntt(fname, x, N.FORWARD, cfg), rf.ntt(x, n, 0, ordering=N.kNR))
    finally:
        lib.destroy_config_extension(ext)
        N.release_domain(fname)
        rf.release_domain()


@pytest.mark.parametrize("victim", [0, 3])
def test_split_ntt_slot_failure_between_two_exchanges(hip, slots, victim):
    """a slot that fails AFTER the first gate of the split transform -- here right before its second all-to-all
    -- must not leave its peers waiting in ncclRecv: every exchange has its own gate, the call returns an error, and the
    same call without the fault succeeds."""
    from icicle_amd import ntt as N
    from icicle_amd._lib import IcicleError, lib, check

    fname, F, logn, P = "babybear", pyref.BABYBEAR, 12, 4
    n = 1 << logn
    rng = np.random.default_rng(victim)
    rf = ref.RefNttField(fname)
    root = N.get_root_of_unity(fname, n)
    rf.init_domain(root)
    N.init_domain(fname, root)
    ext = _ext(hip_num_devices=P)
    try:
        slots(P)
        x = rng.integers(0, F.p, size=n, dtype=np.uint32)
        cfg = hip.NTTConfigU32.default()
        cfg.ext = ext
        check(lib.icicle_hip_test_inject_failure(victim, 2))
        with pytest.raises(IcicleError):
            N.ntt(fname, x, N.FORWARD, cfg)
        assert np.array_equal(N.ntt(fname, x, N.FORWARD, cfg), rf.ntt(x, n, 0))  # one-shot fault: gone
    finally:
        lib.destroy_config_extension(ext)
        N.release_domain(fname)
        rf.release_domain()


def test_two_host_threads_call_the_multi_device_msm_at_once(hip, slots):
    """collectives of two calls on one communicator set must not interleave -- the per-set mutex serialises
    host threads that enter a multi-device msm() on the same devices at the same time (ctypes drops the GIL in the call)."""
    import threading

    from icicle_amd import msm as M
    from icicle_amd._lib import lib

    C, rng, bases, sc = _inputs("bn254", 4001, 4242)
    refc = ref.RefCurve("bn254")
    sc2 = to_words(rand_scalars(rng, len(sc), C.r), 8)
    exp = [refc.to_affine(refc.msm(s, bases)) for s in (sc, sc2)]
    slots(4)
    exts = [_ext(hip_num_devices=4, hip_msm_exchange_buckets=bool(k)) for k in range(2)]
    got, errs = [None, None], []

    def run(k, scalars):
        try:
            for _ in range(4):
                cfg = hip.MSMConfig.default()
                cfg.ext = exts[k]
                got[k] = M.msm("bn254", scalars, bases, cfg)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    try:
        th = [threading.Thread(target=run, args=(0, sc)), threading.Thread(target=run, args=(1, sc2))]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=150)
        assert not errs, errs
        assert all(not t.is_alive() for t in th), "a multi-device call is stuck"
        for k in range(2):
            assert np.array_equal(refc.to_affine(got[k]), exp[k]), k
    finally:
        for e in exts:
            lib.destroy_config_extension(e)


def test_resident_copies_die_with_the_allocation(hip, slots):
    """ "hip_bases_resident" copies are keyed on the base pointer: when the caller frees that device allocation
    (icicle_free), the copies must go with it -- the next allocation may get the same address with other points."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, multi_stats
    from icicle_amd.runtime import DeviceVec

    C, rng, bases, sc = _inputs("bn254", 3001, 5151)
    refc = ref.RefCurve("bn254")
    bases2 = np.ascontiguousarray(bases[::-1])
    slots(4)
    ext = _ext(hip_num_devices=4, hip_bases_resident=True)
    try:
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        d1 = DeviceVec.from_host(bases)
        got = M.msm("bn254", sc, d1, cfg, msm_size=len(sc))
        assert np.array_equal(refc.to_affine(got), refc.to_affine(refc.msm(sc, bases)))
        addr = d1.ptr
        d1.free()
        d2 = DeviceVec.from_host(bases2)  # very likely the same address
        multi_stats(reset=True)
        got2 = M.msm("bn254", sc, d2, cfg, msm_size=len(sc))
        st = multi_stats()
        assert np.array_equal(refc.to_affine(got2), refc.to_affine(refc.msm(sc, bases2))), ("stale resident copy served", d2.ptr == addr)
        assert st["staged_base_bytes"] > 0 and st["resident_base_hits"] == 0, st
    finally:
        lib.destroy_config_extension(ext)


def test_resident_copies_versioned_by_the_callers_generation(hip, slots):
    """memory that does not come from icicle_malloc (here: a host array rewritten IN PLACE) cannot tell the cache
    that its contents changed -- "hip_bases_generation" is part of the key, so a new value stages fresh copies, and the old
    value keeps serving the old ones until they are released."""
    from icicle_amd import msm as M
    from icicle_amd._lib import lib, multi_stats

    C, rng, bases, sc = _inputs("bn254", 2500, 6161)
    refc = ref.RefCurve("bn254")
    buf = bases.copy()
    exp1 = refc.to_affine(refc.msm(sc, buf))
    slots(4)
    ext = _ext(hip_num_devices=4, hip_bases_resident=True, hip_bases_generation=1)
    try:
        cfg = hip.MSMConfig.default()
        cfg.ext = ext
        assert np.array_equal(refc.to_affine(M.msm("bn254", sc, buf, cfg)), exp1)
        buf[:] = bases[::-1]  # same address, other points
        exp2 = refc.to_affine(refc.msm(sc, buf))
        lib.config_extension_set_int(ext, b"hip_bases_generation", 2)
        multi_stats(reset=True)
        assert np.array_equal(refc.to_affine(M.msm("bn254", sc, buf, cfg)), exp2)
        st = multi_stats()
        assert st["staged_base_bytes"] == buf.nbytes and st["resident_base_hits"] == 0, st
        multi_stats(reset=True)
        assert np.array_equal(refc.to_affine(M.msm("bn254", sc, buf, cfg)), exp2)  # generation 2 again: served from the copies
        assert multi_stats()["staged_base_bytes"] == 0
    finally:
        lib.icicle_hip_msm_release_resident_bases(None)
        lib.destroy_config_extension(ext)


def test_operand_copies_when_peer_access_is_refused(hip, slots):
    """the copy route the in-process multi-GPU paths take when hipDeviceEnablePeerAccess is refused (or
    ICICLE_HIP_NO_PEER_ACCESS=1): device-resident operands and results of the slots other than the caller's go through
    hipMemcpyPeerAsync -- which the HIP runtime stages through host memory between non-peer devices -- instead of direct
    hipMemcpyDefault copies (common.h PeerRoute). Rehearsed on 4 slots: MSM (E1 and E2), base residency, batched NTT row
    shards; results must equal the reference and the counter must show that the route was taken."""
    from icicle_amd import msm as M
    from icicle_amd import ntt as N
    from icicle_amd._lib import lib, check, multi_stats
    from icicle_amd.runtime import DeviceVec

    C, rng, bases, sc = _inputs("bn254", 5003, 4242)
    n = len(sc)
    refc = ref.RefCurve("bn254")
    sc2 = np.vstack([sc, to_words(rand_scalars(rng, n, C.r), 8)])
    exp = refc.to_affine(refc.msm(sc2, bases, batch=2, shared=True))
    slots(4)
    check(lib.icicle_hip_test_set_no_peer_access(True))
    d_sc, d_b = DeviceVec.from_host(sc2), DeviceVec.from_host(bases)
    try:
        for exchange in (False, True):
            ext = _ext(hip_num_devices=4, hip_msm_exchange_buckets=exchange)
            try:
                multi_stats(reset=True)
                cfg = hip.MSMConfig.default()
                cfg.ext, cfg.batch_size = ext, 2
                got = M.msm("bn254", d_sc, d_b, cfg, msm_size=n)
                assert np.array_equal(refc.to_affine(got), exp), exchange
                st = multi_stats()
                assert st["threaded_calls"] == 1 and st["peer_staged_copies"] >= 3 * (2 + 1), st  # three remote slots: two scalar rows + the bases each
            finally:
                lib.destroy_config_extension(ext)
        # batched NTT row shards, device-resident rows in and out
        F = pyref.BABYBEAR
        logn, rows = 12, 16
        x = rng.integers(0, F.p, size=rows << logn, dtype=np.uint32)
        N.init_domain("babybear", N.get_root_of_unity("babybear", 1 << logn))
        rf = ref.RefNttField("babybear")
        rf.init_domain(rf.get_root_of_unity(1 << logn))
        ext = _ext(hip_num_devices=4)
        d_x, d_y = DeviceVec.from_host(x), DeviceVec(x.nbytes)
        try:
            multi_stats(reset=True)
            ncfg = hip.NTTConfigU32.default()
            ncfg.batch_size, ncfg.ext = rows, ext
            N.ntt("babybear", d_x, N.FORWARD, ncfg, out=d_y, size=1 << logn)
            assert np.array_equal(d_y.to_host(), rf.ntt(x, 1 << logn, 0, batch=rows))
            assert multi_stats()["peer_staged_copies"] >= 6  # three remote slots, rows in and rows out
        finally:
            lib.destroy_config_extension(ext)
            d_x.free(), d_y.free()
            N.release_domain("babybear")
            rf.release_domain()
    finally:
        check(lib.icicle_hip_test_set_no_peer_access(False))
        d_sc.free(), d_b.free()
