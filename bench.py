#!/usr/bin/env python3
"""bench.py -- BN254 MSM 2^26 (primary, BASELINE.json configs[1]) + BabyBear NTT 2^24 x 64
(secondary, configs[2]) on N MI355X, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N ...      (no WORLD_SIZE in the environment: re-launches itself under torch.distributed.run;
                                     on a box with fewer than N GPUs it prints a JSON line with an "error" key, exit 0)

At N > 1 the line carries a second object, "inproc": the same shard cut driven by ONE process through the C ABI
(MSMConfig.ext / NTTConfig.ext "hip_num_devices" = N, one host thread + stream per GPU inside the call, in-library RCCL
exchange, bases resident per GPU) -- what a Rust / Go / C++ caller of msm() / ntt() reaches. --no-inproc skips it.

A step = one pass of the hot path over one batch of synthetic input already resident in HBM:
one 2^26-term BN254 MSM per rank (weak scaling: the N shards form one 2^26*N-term MSM whose partial
results are all-gathered over RCCL and summed on every rank). Timed region: barrier + synchronize on
both sides, max over ranks. One JSON line on rank 0.

--scaling strong cuts ONE 2^26 MSM over the N GPUs instead (2^26/N pairs per rank, same exchange); the default (weak)
line at N > 1 carries that figure as well, as the object "strong", so one invocation records both.

Extra objects: "roofline" (dominant kernel = MSM bucket accumulation, duration from hipEvents on the
launch stream, algorithmic bytes per SURVEY.md 8(d); its ALU and gather roofs are measured in the same
run), "cpu_baseline" (the reference CPU backend from oracle/_ref timed on this box's host cores -- on
the full workload when the host grants >= 16 cores of CPU time), "ntt" (secondary metric with its own
roofline/cpu_baseline).  --size-log2 / --ntt-log2 shrink the workload for quick checks; the
JSON then names the reduced workload (never reported as the headline config).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 4.63 TB/s measured copy
BN254_R_TOP = 0x30644E72


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-log2", type=int, default=26, help="MSM size per GPU (headline: 26)")
    ap.add_argument("--ntt-log2", type=int, default=24, help="NTT size (headline: 24)")
    ap.add_argument("--ntt-batch", type=int, default=64)
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--msm-c", type=int, default=0, help="force the MSM window size (0 = backend default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-msm-log2", type=int, default=0, help="CPU baseline MSM size; 0 = the full workload when the host grants >= 16 cores, else 2^20")
    ap.add_argument("--cpu-ntt-log2", type=int, default=0, help="CPU baseline NTT size; 0 = the full size when the host grants >= 16 cores, else 2^20")
    ap.add_argument("--no-inproc", action="store_true", help="N > 1: skip the single-process hip_num_devices=N measurement")
    ap.add_argument("--no-shard-extras", action="store_true", help="N = 1: skip the config3_shard / config4_shard objects (per-GPU shares of BASELINE configs[3] and [4])")
    ap.add_argument("--inproc-only", action="store_true", help=argparse.SUPPRESS)  # internal: the launcher's second leg
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: one 2^size MSM per GPU (the N shards form one 2^size*N MSM); strong: ONE 2^size MSM cut over the N GPUs")
    return ap.parse_args()


def synth_scalars(n, device, seed):
    """n x 8 uint32 limbs, value < r (top limb drawn below r's top limb), resident on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    s = torch.empty((n, 8), dtype=torch.int32, device=device)
    step = 1 << 26  # drawn in slices: the in-process N-GPU leg asks for N * 2^26 scalars (2^32 words at N = 8)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        s[lo:hi] = torch.randint(-(2 ** 31), 2 ** 31, (hi - lo, 8), dtype=torch.int32, device=device, generator=g)
        s[lo:hi, 7] = torch.randint(0, BN254_R_TOP, (hi - lo,), dtype=torch.int32, device=device, generator=g)
    return s



REHEARSAL = os.environ.get("ICICLE_BENCH_GLOO_REHEARSAL", "0") == "1"


class GlooViaHost:
    """ICICLE_BENCH_GLOO_REHEARSAL=1: the one-process-per-GPU leg on a box with FEWER GPUs than ranks -- every rank works on
    GPU 0 and the collectives run over gloo on host copies of the tensors. Not a measurement: it exists so that the
    launcher, the shard cut, the weak and strong legs, the split transform, the watchdogs and the JSON assembly have all
    executed with world > 1 before an 8-GPU node runs them for real (tests/test_bench_cli.py)."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def barrier(self):
        self._d.barrier()

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(h)

    def all_gather_into_tensor(self, out, inp):
        torch.cuda.synchronize()
        ho, hi = out.cpu(), inp.cpu()
        self._d.all_gather_into_tensor(ho, hi)
        out.copy_(ho)

    def all_to_all_single(self, recv, send):
        torch.cuda.synchronize()
        hr, hs = recv.cpu(), send.cpu()
        self._d.all_to_all_single(hr, hs)
        recv.copy_(hr)

    def destroy_process_group(self):
        self._d.destroy_process_group()

    def get_world_size(self):
        return self._d.get_world_size()

    def get_backend(self):
        return self._d.get_backend()

    def all_gather_object(self, out, obj):
        self._d.all_gather_object(out, obj)


def effective_host_cores():
    """CPU time the process can really get: the cgroup quota if there is one (the GPU boxes of this build show 256 hardware threads and
    grant 16 cores: cat /sys/fs/cgroup/cpu.max -> 1600000 100000), else the visible cores."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def visible_gpus():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except Exception:
                continue
    return None


def launch_self(args):
    """`python bench.py --gpus N` with no rendezvous in the environment: run the one-process-per-GPU job under
    torch.distributed.run ourselves, then (GPUs free again) the single-process hip_num_devices = N leg, and print ONE line."""
    import subprocess

    have = visible_gpus()
    if have < args.gpus and not (REHEARSAL and have >= 1):
        print(json.dumps({"metric": "bn254_msm_2^26_per_sec", "value": None, "unit": "MSM/s", "n_gpus": args.gpus, "steps": args.steps,
                          "warmup": args.warmup, "higher_is_better": True, "scaling": args.scaling, "data": "synthetic",
                          "error": f"needs {args.gpus} devices, {have} visible"}))
        return
    passthrough = [a for a in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + passthrough + ["--no-inproc"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    except subprocess.TimeoutExpired as e:
        r = subprocess.CompletedProcess(cmd, 124, stdout=(e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""), stderr="one-process-per-GPU leg timed out after 1200 s")
    out = _last_json_line(r.stdout)
    if out is None:
        out = {"metric": "bn254_msm_2^26_per_sec", "value": None, "unit": "MSM/s", "n_gpus": args.gpus,
               "error": f"one-process-per-GPU leg failed (rc {r.returncode}): " + (r.stderr or r.stdout)[-1500:]}
    if not args.no_inproc:
        try:
            r2 = subprocess.run([sys.executable, os.path.abspath(__file__)] + passthrough + ["--inproc-only"], capture_output=True, text=True,
                                env=env, timeout=240)
            out["inproc"] = _last_json_line(r2.stdout) or {"error": f"rc {r2.returncode}: " + (r2.stderr or r2.stdout)[-1500:]}
        except Exception as e:  # the second leg never costs the primary line
            out["inproc"] = {"error": repr(e)}
    print(json.dumps(out))


def _inproc_collectives(lib):
    """what the in-library communicator layer saw (icicle_hip_collectives_info): library version, devices asked for, ncclCommCount,
    rank order verified, communicator sets created"""
    v = (ctypes.c_int * 5)()
    try:
        if lib.icicle_hip_collectives_info(v, 5) != 0:
            return {"error": "icicle_hip_collectives_info failed"}
    except Exception as e:
        return {"error": repr(e)}
    return {"nccl_version_code": v[0], "devices_requested": v[1], "nccl_comm_count": v[2], "ranks_and_count_as_expected": {1: True, 0: False}.get(v[3]),
            "communicator_sets_created": v[4]}


def inproc_main(args):
    """ONE process, N GPUs behind the C ABI: a single msm() / ntt() call with ext "hip_num_devices" = N. Weak scaling like
    the primary line: one MSM of N * 2^size terms whose operands live on GPU 0 (the caller's device); bases are placed on
    their devices by the first call ("hip_bases_resident"), so a timed call ships scalars only. Prints one JSON object."""
    import icicle_amd
    from icicle_amd import msm as M
    from icicle_amd import ntt as N
    from icicle_amd import runtime
    from icicle_amd._lib import MSMConfig, NTTConfigU32, lib, check, multi_stats

    G = args.gpus
    have = runtime.get_device_count()
    rehearsal = os.environ.get("ICICLE_BENCH_INPROC_VIRTUAL", "0") == "1"  # this leg's own code on a box with fewer GPUs: virtual slots + loopback collectives
    if rehearsal:
        check(lib.icicle_hip_test_set_virtual_devices(G), "virtual devices")
        check(lib.icicle_hip_set_collectives_library(os.path.join(ROOT, "tests", "_build", "libnccl_loopback.so").encode()), "loopback collectives")
    elif have < G:
        print(json.dumps({"error": f"needs {G} devices, {have} visible"}))
        return
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    runtime.set_device(0)
    strong = args.scaling == "strong"
    n = (1 << args.size_log2) * (1 if strong else G)
    res = {"form": "one process, MSMConfig.ext / NTTConfig.ext hip_num_devices = N (one host thread + stream per GPU inside the call)",
           "n_gpus": G, "scaling": args.scaling}
    if rehearsal:
        res["rehearsal"] = f"{G} virtual device slots on {have} GPU(s), loopback collectives: timings mean nothing"
    ext = lib.create_config_extension()
    try:
        lib.config_extension_set_int(ext, b"hip_num_devices", G)
        lib.config_extension_set_bool(ext, b"hip_bases_resident", True)
        bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
        check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, 1, True, None), "generate")
        scalars = synth_scalars(n, dev, 4321)
        out = torch.empty(24, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        cfg = MSMConfig.default()
        cfg.c = args.msm_c
        cfg.ext = ext

        def step():
            M.msm("bn254", scalars.data_ptr(), bases.data_ptr(), cfg, results=out.data_ptr(), msm_size=n)

        step()  # places the base shards on their devices
        multi_stats(reset=True)
        for _ in range(max(0, args.warmup - 1)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = multi_stats()
        calls = args.steps + max(0, args.warmup - 1)
        res.update({"metric": "bn254_msm_2^26_per_sec", "unit": "MSM/s", "value": (n / float(1 << 26)) * args.steps / dt,
                    "ms_per_step": dt / args.steps * 1e3, "steps": args.steps,
                    "workload": f"ONE BN254 MSM of {G if not strong else 1} x 2^{args.size_log2} terms, operands on GPU 0, bases resident per GPU after the first call",
                    "wire_bytes_per_call": {"bases": st["staged_base_bytes"] / max(1, calls), "scalars": st["staged_scalar_bytes"] / max(1, calls)},
                    "resident_base_hits_per_call": st["resident_base_hits"] / max(1, calls)})
        # the result of the last timed call, checked: the same sum as G plain device-0 MSMs over the shards (no extension:
        # the single-GPU path every parity test covers), added up together with the NEGATED multi-device result by the
        # library's complete projective sum -- which must come out as the identity (Z = 0). No oracle involved.
        lib.icicle_hip_msm_release_resident_bases(None)
        from icicle_amd import dist as D

        parts = torch.zeros((G + 1, 24), dtype=torch.int32, device=dev)
        plain = MSMConfig.default()
        plain.c = args.msm_c
        for g in range(G):
            lo, hi = D.shard_range(n, g, G)
            M.msm("bn254", scalars[lo:hi].data_ptr(), bases[lo:hi].data_ptr(), plain, results=parts[g].data_ptr(), msm_size=hi - lo)
        torch.cuda.synchronize()
        w = out.cpu().numpy().view(np.uint32).copy()
        BN254_P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
        yv = sum(int(w[8 + i]) << (32 * i) for i in range(8))
        ny = (BN254_P - yv) % BN254_P
        w[8:16] = np.array([(ny >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
        parts[G] = torch.from_numpy(w.view(np.int32)).to(dev)
        tot = torch.empty(24, dtype=torch.int32, device=dev)
        check(lib.bn254_hip_projective_sum(parts.data_ptr(), G + 1, tot.data_ptr(), None), "projective_sum")
        torch.cuda.synchronize()
        th = tot.cpu().numpy().view(np.uint32)
        res["result_ok"] = bool(not th[16:24].any() and w[16:24].any())  # sum - result == identity, and the result itself is a finite point
        del bases, scalars, parts
    except Exception as e:  # the MSM leg's failure keeps the NTT leg below
        res["error"] = "msm leg: " + repr(e)
    res["collectives"] = _inproc_collectives(lib)
    try:
        if not args.no_ntt:
            logn, rows = args.ntt_log2, args.ntt_batch * G
            nn = 1 << logn
            N.init_domain("babybear", N.get_root_of_unity("babybear", nn))
            g = torch.Generator(device=dev)
            g.manual_seed(77)
            x = torch.empty((rows, nn), dtype=torch.int32, device=dev)
            for r0 in range(0, rows, 16):
                x[r0:r0 + 16] = torch.randint(0, 0x78000001, (min(16, rows - r0), nn), dtype=torch.int32, device=dev, generator=g)
            y = torch.empty_like(x)
            z = torch.empty_like(x)
            ncfg = NTTConfigU32.default()
            ncfg.batch_size = rows
            ext2 = lib.create_config_extension()
            lib.config_extension_set_int(ext2, b"hip_num_devices", 4 * G)  # 4 row shards per GPU: upload / compute / download overlap
            ncfg.ext = ext2

            def nstep():
                N.ntt("babybear", x.data_ptr(), N.FORWARD, ncfg, out=y.data_ptr(), size=nn)
                N.ntt("babybear", y.data_ptr(), N.INVERSE, ncfg, out=z.data_ptr(), size=nn)

            nstep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                nstep()
            torch.cuda.synchronize()
            dtn = time.perf_counter() - t0
            res["ntt"] = {"metric": f"babybear_ntt_2^{logn}_per_sec", "unit": "NTT/s", "value": 2 * rows * args.steps / dtn,
                          "ms_per_step": dtn / args.steps * 1e3, "roundtrip_ok": bool(torch.equal(x, z)),
                          "workload": f"BabyBear NTT 2^{logn}, batch {rows} on GPU 0, forward + inverse, 4 row shards per GPU"}
            lib.destroy_config_extension(ext2)
            N.release_domain("babybear")
    except Exception as e:
        res["ntt"] = {"error": repr(e)}
    finally:
        lib.destroy_config_extension(ext)
    res["collectives_after_ntt"] = _inproc_collectives(lib)
    print(json.dumps(res))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.inproc_only:
        return inproc_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_self(args)
    if world != args.gpus:
        print(json.dumps({"metric": "bn254_msm_2^26_per_sec", "value": None, "unit": "MSM/s", "n_gpus": args.gpus,
                          "error": f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"}))
        return
    if REHEARSAL:
        local_rank = 0  # every rank on GPU 0, collectives over gloo (GlooViaHost)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        if REHEARSAL:
            dist.init_process_group("gloo")
            dist = GlooViaHost(dist)
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    import icicle_amd
    from icicle_amd import dist as D
    from icicle_amd import msm as M
    from icicle_amd import ntt as N
    from icicle_amd import runtime
    from icicle_amd._lib import MSMConfig, NTTConfigU32, lib, check

    runtime.set_device(local_rank)

    def barrier_sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---------------- MSM: synthetic inputs resident in HBM ----------------
    strong = args.scaling == "strong"
    if strong:  # ONE 2^size_log2 MSM; rank r holds pairs [lo, hi) of it (north_star: "large MSMs shard bases across the GPUs")
        lo, hi = D.shard_range(1 << args.size_log2, rank, world)
        n, k0 = hi - lo, 1 + lo
    else:
        n, k0 = 1 << args.size_log2, 1 + rank * (1 << 40)
    bases = torch.empty((n, 16), dtype=torch.int32, device=dev)
    # distinct points (k0 + i)G, a different range per rank; generated on the GPU
    check(lib.bn254_hip_generate_affine_points(bases.data_ptr(), n, k0, True, None), "generate")
    scalars = synth_scalars(n, dev, 1234 + rank)
    torch.cuda.synchronize()

    def msm_step():
        cfg = MSMConfig.default()
        cfg.c = args.msm_c
        return D.msm_sharded("bn254", scalars, bases, n, rank, world, dist, cfg)

    for _ in range(args.warmup):
        msm_step()
    lib.icicle_hip_enable_kernel_timing(True)
    tot, cnt = ctypes.c_double(), ctypes.c_int()
    for which in (0, 2, 3):
        lib.icicle_hip_kernel_timing(which, True, ctypes.byref(tot), ctypes.byref(cnt))
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = msm_step()
    barrier_sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    lib.icicle_hip_kernel_timing(0, True, ctypes.byref(tot), ctypes.byref(cnt))
    lib.icicle_hip_enable_kernel_timing(False)
    msm_ms = dt / args.steps * 1e3
    # ---- what the collectives layer saw: recorded in the line, and the gathered result must be the SAME bytes
    # on every rank (each rank sums the all-gathered partials itself: same inputs, same deterministic kernel)
    collectives = {"world_size": world, "backend": None, "rccl_version": None}
    try:
        collectives["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:
        collectives["rccl_version"] = f"unavailable: {e!r}"
    if world > 1:
        try:
            collectives["world_size"] = dist.get_world_size()
            collectives["backend"] = dist.get_backend()
            mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(),
                    "icicle_device": runtime.get_active_device() if hasattr(runtime, "get_active_device") else None}
            seen = [None] * world
            dist.all_gather_object(seen, mine)
            collectives["ranks"] = seen
            allres = torch.empty((world, res.numel()), dtype=res.dtype, device=dev)
            dist.all_gather_into_tensor(allres.reshape(-1), res.contiguous().reshape(-1))
            same = torch.tensor([1.0 if bool((allres == res.reshape(1, -1)).all()) else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            collectives["result_identical_on_all_ranks"] = bool(same.item() == 1.0)
        except Exception as e:
            collectives["error"] = repr(e)
    # bucket accumulation of ONE MSM = the sum of its k_accumulate launches (one per window group of the pipelined schedule)
    acc_ms = tot.value / max(1, args.steps)
    acc_launches = cnt.value
    phases = {}
    for which, name in ((2, "sort_exposed_ms"), (3, "tail_exposed_ms")):
        lib.icicle_hip_kernel_timing(which, True, ctypes.byref(tot), ctypes.byref(cnt))
        phases[name] = tot.value / max(1, args.steps)
    msm_bytes = n * (32 + 64) + 96  # N*sizeof(scalar) + N*sizeof(affine) + sizeof(projective)
    total_terms = (1 << args.size_log2) if strong else world * n
    units_per_step = total_terms / float(1 << 26)  # in 2^26-term MSMs
    value = units_per_step * args.steps / dt
    roofline = {
        "bound": "hbm", "kernel": "k_accumulate<bn254_g1>", "achieved": msm_bytes / (acc_ms * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": msm_bytes / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": None, "avg_launch_ms": acc_ms, "launches": acc_launches, "launches_per_msm": acc_launches / max(1, args.steps),
        "phases_ms": {"accumulate": acc_ms, **phases, "whole_msm": msm_ms},
        "note": "MSM is integer-ALU bound (v_mad_u64_u32), not HBM bound; see DESIGN.md and 'alu'",
    }
    # HBM traffic of the dominant kernel from the committed PMC profile of this same workload (bench.py cannot
    # run rocprofv3 on itself); only attached when the workload matches the profiled one. FETCH_SIZE is calibrated on
    # a kernel with the same access pattern and a known byte count (icicle_hip_ubench_gather under the same counter).
    # The PMC file carries the SHA-256 of the kernel sources it was taken on (tools/pmc_json.py); the traffic is attached only
    # when the sources THIS run executes hash the same -- a counter value of another build says nothing about this one.
    pmc = {}
    try:
        import glob

        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))[-1]
        with open(newest) as f:
            pmc = json.load(f)
        pmc["_file"] = os.path.basename(newest)
    except Exception:
        pmc = {}

    def sources_sha16(names):
        import hashlib

        h = hashlib.sha256()
        try:
            for nm in names:
                with open(os.path.join(ROOT, "icicle_amd", "csrc", nm), "rb") as f:
                    h.update(f.read())
        except Exception:
            return None
        return h.hexdigest()[:16]

    msm_sha = sources_sha16(["msm_impl.hpp", "msm_plan.h", "ec.hpp", "bigfield.hpp", "mont_asm.hpp"])
    ntt_sha = sources_sha16(["ntt_fast.hpp", "ntt.hip", "ntt_plan.h", "smallfield.hpp"])
    if args.size_log2 == 26 and args.msm_c == 0 and not strong and "msm_bn254_2^26" in pmc:
        if pmc.get("msm_sources_sha16") == msm_sha and msm_sha:
            m = pmc["msm_bn254_2^26"]
            roofline["traffic"] = (m["fetch_size_kb_raw"] * m["fetch_correction"] + m["write_size_kb"]) * 1024 / 1e9
            roofline["traffic_unit"] = f"GB per launch (FETCH_SIZE x calibration + WRITE_SIZE, profiles/{pmc['_file']}, kernel sources {msm_sha})"
        else:
            roofline["traffic_note"] = f"profiles/{pmc.get('_file')} was taken on other kernel sources ({pmc.get('msm_sources_sha16')} != {msm_sha}): not attached"
    # secondary: integer-ALU view. mixed adds per MSM = n * windows; 8M + 2S field operations each. The roof is
    # measured in this run: the same ec.hpp mixed add with every operand in registers (no memory traffic at all).
    pc, pw = ctypes.c_int(), ctypes.c_int()
    pcfg = MSMConfig.default()
    pcfg.c = args.msm_c
    check(lib.icicle_hip_msm_plan(n, 254, ctypes.byref(pcfg), ctypes.byref(pc), ctypes.byref(pw)), "msm_plan")
    madds = n * pw.value
    rate = ctypes.c_double()
    check(lib.icicle_hip_ubench_mixed_add(0, ctypes.byref(rate)), "ubench_mixed_add")
    roofline["alu"] = {"window_bits": pc.value, "windows": pw.value, "mixed_adds": madds,
                       "mixed_adds_per_s": madds / (acc_ms * 1e-3), "roof": rate.value,
                       "frac": madds / (acc_ms * 1e-3) / rate.value,
                       "roof_is": "XYZZ mixed adds/s, operands in registers, measured in this run (icicle_hip_ubench_mixed_add)",
                       # the hardware's own figure beside the self-referential one: pure v_mad_u64_u32 issue. 1024 SIMDs x f / 6.05
                       # cycles per wave-instruction x 64 lanes / 1467 mads per mixed add (profiles/r02_alu_ubench.txt); f = 2.4 GHz
                       # nominal. Under this all-mad load the chip sustains ~0.83 of the clock it holds half-loaded
                       # (profiles/r03_notes.md section 1), i.e. ~14.7e9/s.
                       "hw_roof_nominal_clock": 1024 * 2.4e9 / 6.05 * 64 / 1467,
                       "hw_frac_nominal_clock": madds / (acc_ms * 1e-3) / (1024 * 2.4e9 / 6.05 * 64 / 1467),
                       "hw_roof_is": "v_mad_u64_u32 issue roof: 1024 SIMDs x 2.4 GHz / 6.05 cycles x 64 lanes / 1467 mads per mixed add"}
    # every mixed add gathers one 64-byte point at a random address of the bases array: ceiling of that access pattern
    # over a region of the same size, measured in this run
    check(lib.icicle_hip_ubench_gather(max(64, n * 64), 1 << 27, ctypes.byref(rate)), "ubench_gather")
    roofline["gather"] = {"gathers_per_s": madds / (acc_ms * 1e-3), "measured_ceiling": rate.value,
                          "frac": madds / (acc_ms * 1e-3) / rate.value,
                          "unit": f"random 64-byte gathers/s over {n * 64 / 2**30:.2f} GiB, measured in this run"}
    check(lib.icicle_hip_release_workspace(), "release_workspace")

    out = {
        "metric": "bn254_msm_2^26_per_sec", "value": value, "unit": "MSM/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": msm_ms, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u32x8 (254-bit integer field, 29-bit-limb Montgomery)", "data": "synthetic",
        "config": {"workload": (f"BN254 G1 MSM, ONE 2^{args.size_log2}-term MSM cut over {world} GPU(s), " if strong else
                                f"BN254 G1 MSM, 2^{args.size_log2} uniform scalars x 2^{args.size_log2} distinct affine bases per GPU, ")
                               + "batch 1, inputs resident in HBM, precompute_factor 1",
                   "sharding": "bases/scalars sharded per rank; RCCL all_gather of partial sums + projective add"},
        "roofline": roofline,
        "collectives": collectives,
    }
    if world > 1 and collectives.get("result_identical_on_all_ranks") is False:
        out["error"] = "the all-gathered MSM result differs between ranks"
    if REHEARSAL:
        out["rehearsal"] = f"{world} ranks sharing GPU 0, collectives over gloo on host copies: timings mean nothing"

    # ---------------- N > 1, default (weak) line: the STRONG figure in the same invocation ----------------
    # ONE 2^size MSM cut over the N GPUs (2^size / N pairs per rank, a prefix of this rank's resident inputs; same exchange),
    # so that a single driver run records both scaling modes. Never costs the primary line.
    if world > 1 and not strong and os.environ.get("ICICLE_BENCH_STRONG", "1") == "1":
        import threading

        strong_done = threading.Event()

        def strong_emergency():  # a collective that stalls must not cost the primary line (same guard as the ntt_split object below)
            if strong_done.is_set():
                return
            out["strong"] = {"error": "no completion within 120 s; skipped"}
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)

        strong_watchdog = threading.Timer(120.0, strong_emergency)
        strong_watchdog.daemon = True
        strong_watchdog.start()
        try:
            slo, shi = D.shard_range(1 << args.size_log2, rank, world)
            ns = shi - slo

            def strong_step():
                cfg = MSMConfig.default()
                cfg.c = args.msm_c
                return D.msm_sharded("bn254", scalars[:ns], bases[:ns], ns, rank, world, dist, cfg)

            strong_step()
            barrier_sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                strong_step()
            barrier_sync()
            dts = max_over_ranks(time.perf_counter() - t0)
            out["strong"] = {"metric": "bn254_msm_2^26_per_sec", "scaling": "strong", "unit": "MSM/s", "n_gpus": world,
                             "value": ((1 << args.size_log2) / float(1 << 26)) * args.steps / dts, "ms_per_step": dts / args.steps * 1e3, "steps": args.steps,
                             "workload": f"ONE BN254 G1 MSM of 2^{args.size_log2} terms cut over {world} GPUs ({ns} pairs on rank 0), partial sums all-gathered over RCCL"}
        except Exception as e:
            out["strong"] = {"error": repr(e)}
        finally:
            strong_done.set()
            strong_watchdog.cancel()

    # ---------------- NTT secondary ----------------
    if not args.no_ntt:
        logn, batch = args.ntt_log2, args.ntt_batch
        lo, hi = D.ntt_batch_shard(batch * world, rank, world)  # weak scaling: `batch` rows per GPU
        rows = hi - lo
        nn = 1 << logn
        root = N.get_root_of_unity("babybear", nn)
        N.init_domain("babybear", root)
        g = torch.Generator(device=dev)
        g.manual_seed(99 + rank)
        x = torch.randint(0, 0x78000001, (rows, nn), dtype=torch.int32, device=dev, generator=g)
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        cfg = NTTConfigU32.default()
        cfg.batch_size = rows
        cfg.is_async = True

        def ntt_step():
            N.ntt("babybear", x.data_ptr(), N.FORWARD, cfg, out=y.data_ptr(), size=nn)
            N.ntt("babybear", y.data_ptr(), N.INVERSE, cfg, out=z.data_ptr(), size=nn)

        # Warm-up: at least 4 round trips, run back to back right up to the timed region. After ANY idle gap (a host
        # sync followed by a first-use torch kernel such as the equality check below is enough) the first three
        # 2^24 x 64 transforms run 16 % / 11 % / 5 % slower than steady state while the memory-side clocks ramp
        # (profiles/r02_notes.md section 10): a 6-call timed window right behind such a gap reads 5 % high.
        ntt_warm = max(4, args.warmup)
        lib.icicle_hip_enable_kernel_timing(True)
        for _ in range(ntt_warm):
            ntt_step()
        lib.icicle_hip_kernel_timing(1, True, ctypes.byref(tot), ctypes.byref(cnt))
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ntt_step()
        barrier_sync()
        dtn = max_over_ranks(time.perf_counter() - t0)
        lib.icicle_hip_kernel_timing(1, True, ctypes.byref(tot), ctypes.byref(cnt))
        lib.icicle_hip_enable_kernel_timing(False)
        roundtrip_ok = bool(torch.equal(x, z))  # after the timed region: z is the inverse of the last forward transform
        ntt_call_ms = tot.value / max(1, cnt.value)  # one direction, `rows` transforms
        ntt_bytes = 2 * rows * nn * 4  # 2*batch*N*sizeof(elem) per direction
        ntts = 2 * rows * world * args.steps  # forward + inverse each count
        out["ntt"] = {
            "metric": f"babybear_ntt_2^{logn}_per_sec", "value": ntts / dtn, "unit": "NTT/s",
            "ms_per_step": dtn / args.steps * 1e3, "steps": args.steps, "warmup": ntt_warm, "roundtrip_ok": roundtrip_ok,
            "config": {"workload": f"BabyBear NTT 2^{logn}, batch {rows} per GPU, kNN, forward + inverse round trip, "
                                   f"device resident", "sharding": "rows of the batch per rank, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_ntt_fast<babybear> (the 3 pass launches of one direction)",
                         "achieved": ntt_bytes / (ntt_call_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ntt_bytes / (ntt_call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": ntt_call_ms, "launches": cnt.value},
        }
        # VALU-issue roof beside the HBM fraction: the butterfly arithmetic of the passes with every operand in
        # registers, measured in this run. Unit: 16 elements through one 8-stage pass (2 radix-16 register rounds + 16 products with
        # the factor behind the pass); a direction is rows * N * log2(N) / 128 of them.
        try:
            rate = ctypes.c_double()
            check(lib.icicle_hip_ubench_ntt_pass(0, ctypes.byref(rate)), "ubench_ntt_pass")
            units = rows * nn * logn / 128.0
            alu_frac = units / (ntt_call_ms * 1e-3) / rate.value
            hbm_copy_frac = ntt_bytes * ((logn + 7) // 8) / (ntt_call_ms * 1e-3) / 1e9 / 6290.0
            out["ntt"]["roofline"]["alu"] = {
                "pass_units": units, "pass_units_per_s": units / (ntt_call_ms * 1e-3), "roof": rate.value, "frac": alu_frac,
                "roof_ms_per_direction": units / rate.value * 1e3,
                "roof_is": "16-element x 8-stage pass units/s, operands in registers, measured in this run (icicle_hip_ubench_ntt_pass: k_ntt_fast's own ntt_stages code)",
                "hbm_passes": (logn + 7) // 8, "hbm_traffic_frac_of_measured_copy": hbm_copy_frac,
                "binds": "alu" if alu_frac >= hbm_copy_frac else "hbm",
                "binds_note": "frac = share of the VALU-issue roof; hbm_traffic_frac_of_measured_copy = (passes x algorithmic bytes) / time / 6.29 TB/s (the guide's measured float4 copy)"}
        except Exception as e:
            out["ntt"]["roofline"]["alu"] = {"error": repr(e)}
        if logn == 24 and rows == 64 and "ntt_babybear_2^24x64_one_direction" in pmc:
            if pmc.get("ntt_sources_sha16") == ntt_sha and ntt_sha:
                m = pmc["ntt_babybear_2^24x64_one_direction"]
                out["ntt"]["roofline"]["traffic"] = m["passes"] * (m["fetch_size_kb_raw_per_pass"] * m["fetch_correction"] + m["write_size_kb_per_pass"]) * 1024 / 1e9
                out["ntt"]["roofline"]["traffic_unit"] = f"GB per direction (3 pass launches; FETCH_SIZE calibrated on the 4 GiB each pass provably reads, WRITE_SIZE exact; profiles/{pmc['_file']}, kernel sources {ntt_sha})"
            else:
                out["ntt"]["roofline"]["traffic_note"] = f"profiles/{pmc.get('_file')} was taken on other kernel sources ({pmc.get('ntt_sources_sha16')} != {ntt_sha}): not attached"
        N.release_domain("babybear")

    # ---------------- N > 1 only: ONE large NTT split over the ranks (4-step, all-to-all over RCCL/xGMI) -----
    # Guarded twice: an exception only costs this extra object, and a collective that stalls (nothing a single-GPU
    # box can rehearse) trips a watchdog that prints the primary line without it. ICICLE_BENCH_NTT_SPLIT=0 skips it.
    if world > 1 and not args.no_ntt and os.environ.get("ICICLE_BENCH_NTT_SPLIT", "1") == "1":
        import threading

        split_done = threading.Event()

        def emergency():
            if split_done.is_set():
                return
            out["ntt_split"] = {"error": "no completion within 150 s; skipped"}
            if rank == 0:
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(150.0, emergency)
        watchdog.daemon = True
        watchdog.start()
        try:
            slog = 26 if not REHEARSAL else 20
            N.init_domain("babybear", N.get_root_of_unity("babybear", 1 << slog))
            gsp = torch.Generator(device=dev)
            gsp.manual_seed(5 + rank)
            chunk = torch.randint(0, 0x78000001, ((1 << slog) // world,), dtype=torch.int32, device=dev, generator=gsp)
            fwd = D.ntt_distributed("babybear", chunk, slog, False, rank, world, dist)
            back = D.ntt_distributed("babybear", fwd, slog, True, rank, world, dist)
            ok = bool(torch.equal(back, chunk))
            barrier_sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fwd = D.ntt_distributed("babybear", chunk, slog, False, rank, world, dist)
            barrier_sync()
            dts = max_over_ranks(time.perf_counter() - t0)
            out["ntt_split"] = {"metric": f"babybear_single_ntt_2^{slog}_split_over_{world}_gpus_per_sec", "value": args.steps / dts,
                                "unit": "NTT/s", "ms_per_step": dts / args.steps * 1e3, "roundtrip_ok": ok,
                                "exchange": "3 x all_to_all_single (natural-order output), RCCL"}
            N.release_domain("babybear")
        except Exception as e:  # never lose the primary line to the secondary experiment
            out["ntt_split"] = {"error": repr(e)}
        finally:
            split_done.set()
            watchdog.cancel()

    # ---------------- N = 1: the per-GPU shares of BASELINE configs[3] and [4] ----------------
    # configs[3] = BLS12-381 MSM 2^28 over 8 GPUs -> 2^25 terms per GPU; configs[4] = KoalaBear NTT 2^22 x 1024 over 8 GPUs -> 128 rows
    # per GPU. Same measurement as the headline objects (hipEvents on the launch stream, roofs measured in this run); reproducible
    # from profiles/r06_config3_shard_* / r06_config4_shard_*. Never costs the primary line.
    if world == 1 and not args.no_shard_extras and args.size_log2 == 26 and args.ntt_log2 == 24:
        try:
            n3 = 1 << 25
            b3 = torch.empty((n3, 24), dtype=torch.int32, device=dev)
            check(lib.bls12_381_hip_generate_affine_points(b3.data_ptr(), n3, 3, True, None), "generate")
            g3 = torch.Generator(device=dev)
            g3.manual_seed(381)
            s3 = torch.randint(-(2 ** 31), 2 ** 31, (n3, 8), dtype=torch.int32, device=dev, generator=g3)
            s3[:, 7] = torch.randint(0, 0x73EDA753, (n3,), dtype=torch.int32, device=dev, generator=g3)
            r3 = torch.empty(36, dtype=torch.int32, device=dev)
            c3 = MSMConfig.default()
            c3.is_async = True

            def step3():
                M.msm("bls12_381", s3.data_ptr(), b3.data_ptr(), c3, results=r3.data_ptr(), msm_size=n3)

            step3()
            lib.icicle_hip_enable_kernel_timing(True)
            for which in (0, 2, 3):
                lib.icicle_hip_kernel_timing(which, True, ctypes.byref(tot), ctypes.byref(cnt))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step3()
            torch.cuda.synchronize()
            dt3 = (time.perf_counter() - t0) / args.steps * 1e3
            ph3 = {}
            for which, name in ((0, "accumulate"), (2, "sort_exposed_ms"), (3, "tail_exposed_ms")):
                lib.icicle_hip_kernel_timing(which, True, ctypes.byref(tot), ctypes.byref(cnt))
                ph3[name] = tot.value / max(1, args.steps)
            lib.icicle_hip_enable_kernel_timing(False)
            check(lib.icicle_hip_msm_plan(n3, 255, ctypes.byref(c3), ctypes.byref(pc), ctypes.byref(pw)), "msm_plan")
            check(lib.icicle_hip_ubench_mixed_add(1, ctypes.byref(rate)), "ubench_mixed_add")
            bytes3 = n3 * (32 + 96) + 144
            madds3 = n3 * pw.value
            out["config3_shard"] = {
                "workload": "BLS12-381 G1 MSM, 2^25 terms = one GPU's share of BASELINE configs[3] (2^28 over 8 GPUs), inputs resident in HBM",
                "ms_per_msm": dt3, "steps": args.steps, "phases_ms": {**ph3, "whole_msm": dt3},
                "kernel": "k_accumulate<bls12_381_g1>", "hbm": {"algorithmic_bytes": bytes3, "achieved_GBps": bytes3 / (ph3["accumulate"] * 1e-3) / 1e9, "frac": bytes3 / (ph3["accumulate"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "alu": {"window_bits": pc.value, "windows": pw.value, "mixed_adds": madds3, "mixed_adds_per_s": madds3 / (ph3["accumulate"] * 1e-3), "roof": rate.value,
                        "frac": madds3 / (ph3["accumulate"] * 1e-3) / rate.value, "roof_is": "BLS12-381 XYZZ mixed adds/s, operands in registers, measured in this run (icicle_hip_ubench_mixed_add(1))"}}
            del b3, s3, r3
            torch.cuda.empty_cache()
            check(lib.icicle_hip_release_workspace(), "release_workspace")
        except Exception as e:
            out["config3_shard"] = {"error": repr(e)}
        try:
            l4, rows4 = 22, 128
            n4 = 1 << l4
            N.init_domain("koalabear", N.get_root_of_unity("koalabear", n4))
            g4 = torch.Generator(device=dev)
            g4.manual_seed(422)
            x4 = torch.randint(0, 0x7F000001, (rows4, n4), dtype=torch.int32, device=dev, generator=g4)
            y4, z4 = torch.empty_like(x4), torch.empty_like(x4)
            c4 = NTTConfigU32.default()
            c4.batch_size, c4.is_async = rows4, True

            def step4():
                N.ntt("koalabear", x4.data_ptr(), N.FORWARD, c4, out=y4.data_ptr(), size=n4)
                N.ntt("koalabear", y4.data_ptr(), N.INVERSE, c4, out=z4.data_ptr(), size=n4)

            lib.icicle_hip_enable_kernel_timing(True)
            for _ in range(4):
                step4()
            lib.icicle_hip_kernel_timing(1, True, ctypes.byref(tot), ctypes.byref(cnt))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step4()
            torch.cuda.synchronize()
            dt4 = (time.perf_counter() - t0) / args.steps * 1e3
            lib.icicle_hip_kernel_timing(1, True, ctypes.byref(tot), ctypes.byref(cnt))
            lib.icicle_hip_enable_kernel_timing(False)
            call4 = tot.value / max(1, cnt.value)
            bytes4 = 2 * rows4 * n4 * 4
            check(lib.icicle_hip_ubench_ntt_pass(1, ctypes.byref(rate)), "ubench_ntt_pass")
            units4 = rows4 * n4 * l4 / 128.0
            out["config4_shard"] = {
                "workload": "KoalaBear NTT 2^22 x 128 rows = one GPU's share of BASELINE configs[4] (1024 rows over 8 GPUs), kNN forward + inverse, device resident",
                "ms_per_round_trip": dt4, "ms_per_direction": call4, "ntt_per_s": 2 * rows4 / (dt4 * 1e-3), "roundtrip_ok": bool(torch.equal(x4, z4)), "steps": args.steps,
                "kernel": "k_ntt_fast<koalabear> (the 3 pass launches of one direction)",
                "hbm": {"algorithmic_bytes": bytes4, "achieved_GBps": bytes4 / (call4 * 1e-3) / 1e9, "frac": bytes4 / (call4 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "alu": {"pass_units": units4, "pass_units_per_s": units4 / (call4 * 1e-3), "roof": rate.value, "frac": units4 / (call4 * 1e-3) / rate.value,
                        "roof_is": "16-element x 8-stage pass units/s, operands in registers, measured in this run (icicle_hip_ubench_ntt_pass(1))"}}
            del x4, y4, z4
            N.release_domain("koalabear")
        except Exception as e:
            out["config4_shard"] = {"error": repr(e)}

    # ---------------- CPU baseline: the reference CPU backend on this box's host cores ----------------
    # On a host that grants >= 16 cores of CPU time (the GPU boxes: 256 visible threads, a cgroup quota of 16 cores) the baseline is timed
    # on the REAL workload: the full 2^26 MSM (about half a minute) and full-size NTT rows; on small hosts a 2^20 sample is timed and the
    # JSON says so.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import ref

            threads_visible = os.cpu_count()
            cores = effective_host_cores()  # what the cgroup grants: the honest "cores" of this baseline
            big_host = cores >= 16
            # the Taskflow stand-in's pool is bounded by twice that (the reference asks for one thread per VISIBLE hardware thread)
            os.environ.setdefault("ICICLE_TASKFLOW_SHIM_MAX_THREADS", str(2 * cores))
            refc = ref.RefCurve("bn254")
            clog = args.cpu_msm_log2 or (args.size_log2 if big_host else min(20, args.size_log2))
            cn = min(n, 1 << clog)
            hb = np.ascontiguousarray(bases[:cn].cpu().numpy().view(np.uint32))
            hs = np.ascontiguousarray(scalars[:cn].cpu().numpy().view(np.uint32))
            # the reference's worker count: its default is one per visible hardware thread, which is NOT its best -- every worker owns and
            # merges a full bucket set, and the GPU boxes grant 16 cores of CPU time for their 256 visible threads (profiles/
            # r06_ref_scaling.txt: BN254 2^24, 32 workers 6.6 s, 256 workers 8.5 s). Timed at the better setting: 2 workers per granted core.
            ref_threads = 2 * cores if cores < threads_visible else 0
            t0 = time.perf_counter()
            exp = refc.msm(hs, hb, n_threads=ref_threads)
            tc = time.perf_counter() - t0
            # parity on the very inputs the CPU was timed on (the full bench inputs when cn == n)
            got = res.cpu().numpy().view(np.uint32).reshape(1, -1) if cn == n else M.msm("bn254", hs, hb)
            parity = bool(np.array_equal(refc.to_affine(got), refc.to_affine(exp)))
            full = cn == n and args.size_log2 == 26
            out["cpu_baseline"] = {
                "value": (cn / float(1 << 26)) / tc, "unit": "MSM/s", "cores": cores, "worker_threads": ref_threads or threads_visible, "host_threads_visible": threads_visible, "kind": "reference",
                "sample": (f"the bench workload itself: one BN254 MSM of 2^{args.size_log2} terms took {tc:.2f} s on the reference CPU backend "
                           f"(oracle/_ref, Taskflow shim, {ref_threads or threads_visible} worker threads on the {cores} cores of CPU time the host grants; {threads_visible} hardware threads are visible)" if cn == n else
                           f"one BN254 MSM of 2^{clog} terms (a prefix of the bench inputs) took {tc:.2f} s on the reference CPU backend; value = "
                           f"that rate in 2^26-term MSMs/s assuming linear scaling (Pippenger is sub-linear per point, so this UNDERSTATES the CPU)"),
                "timed_on_full_workload": full, "parity_with_gpu_on_sample": parity,
            }
            del hb, hs
            if not args.no_ntt:
                rf = ref.RefNttField("babybear")
                cl = args.cpu_ntt_log2 or (args.ntt_log2 if big_host else min(20, args.ntt_log2))
                rf.init_domain(rf.get_root_of_unity(1 << cl))
                crow = min(rows, 4)
                hx = np.ascontiguousarray(x[:crow, : 1 << cl].cpu().numpy().view(np.uint32)).reshape(-1)
                t0 = time.perf_counter()
                cy = rf.ntt(hx, 1 << cl, 0, batch=crow)
                tn = time.perf_counter() - t0
                if tn * rows / crow < 40.0 and crow < rows and cl == args.ntt_log2:  # affordable: time the whole batch
                    crow = rows
                    hx = np.ascontiguousarray(x.cpu().numpy().view(np.uint32)).reshape(-1)
                    t0 = time.perf_counter()
                    cy = rf.ntt(hx, 1 << cl, 0, batch=crow)
                    tn = time.perf_counter() - t0
                scale = 1.0 if cl == args.ntt_log2 else ((1 << cl) * cl) / ((1 << args.ntt_log2) * args.ntt_log2)
                ntt_parity = None
                if cl == args.ntt_log2:
                    ntt_parity = bool(np.array_equal(cy.reshape(crow, -1), y[:crow].cpu().numpy().view(np.uint32)))
                out["ntt"]["cpu_baseline"] = {
                    "value": crow / tn * scale, "unit": "NTT/s", "cores": cores, "kind": "reference",
                    "sample": (f"{crow} forward BabyBear NTTs of 2^{cl} (rows of the bench input) took {tn:.2f} s on the reference CPU backend"
                               + ("" if cl == args.ntt_log2 else f"; scaled by N log N to 2^{args.ntt_log2}")),
                    "timed_on_full_size": cl == args.ntt_log2, "parity_with_gpu_on_sample": ntt_parity,
                }
                rf.release_domain()
        except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
            out.setdefault("cpu_baseline", {"value": None, "unit": "MSM/s", "cores": os.cpu_count(), "kind": "reference",
                                            "sample": f"failed: {e!r}"})

    # ---------------- N > 1 under an external launcher: the single-process hip_num_devices = N leg ----------------
    # (when bench.py launched the ranks itself, launch_self() runs this leg after they have exited.) Every rank gives
    # its device memory back first; rank 0 then runs the leg in a child process with the launcher's variables removed.
    if world > 1 and not args.no_inproc:
        try:
            del scalars, bases
            if not args.no_ntt:
                del x, y, z
            torch.cuda.empty_cache()
            lib.icicle_hip_release_workspace()
            barrier_sync()
            if rank == 0:
                import subprocess

                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                        "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                        "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_NAME")}
                r2 = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--inproc-only"], capture_output=True,
                                    text=True, env=env, timeout=240)
                out["inproc"] = _last_json_line(r2.stdout) or {"error": f"rc {r2.returncode}: " + (r2.stderr or r2.stdout)[-1500:]}
        except Exception as e:
            out["inproc"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
