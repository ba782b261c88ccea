"""CPU: bench.py's launcher contract (VERDICT r02 item 2 i): `python bench.py --gpus N` without a rendezvous in the
environment must not die on an assertion -- it launches itself one rank per GPU, or, on a box with fewer than N GPUs,
prints ONE JSON line with an "error" key and exits 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_without_enough_devices_reports_json_error():
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = max(2, have + 1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--size-log2", "16"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == want and out["value"] is None
    assert f"needs {want} devices" in out["error"]
    assert out["metric"] == "bn254_msm_2^26_per_sec" and out["unit"] == "MSM/s"


def test_bench_rank_count_mismatch_is_a_json_error_too():
    """WORLD_SIZE set by a launcher but different from --gpus: a JSON line, not a traceback"""
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert "error" in out and "WORLD_SIZE=3" in out["error"]


@pytest.mark.gpu
def test_bench_inprocess_leg_rehearsed_on_virtual_slots():
    """the single-process leg of `bench.py --gpus N` (ONE msm() / ntt() call with hip_num_devices = N, bases resident) run
    with 4 virtual device slots + loopback collectives on the one GPU of the test box: the leg's own code, its JSON and
    its wire-byte accounting (the second call ships scalars only)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ICICLE_BENCH_INPROC_VIRTUAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--inproc-only", "--size-log2", "16", "--ntt-log2", "14", "--ntt-batch", "4"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert "error" not in out, out
    assert out["n_gpus"] == 4 and out["value"] > 0 and "rehearsal" in out
    assert out["wire_bytes_per_call"]["bases"] == 0 and out["wire_bytes_per_call"]["scalars"] == 3 * (1 << 16) * 32  # three remote slots
    assert out["resident_base_hits_per_call"] == 3
    assert out["ntt"]["roundtrip_ok"] is True
    assert out["result_ok"] is True  # the timed result equals the sum of plain single-GPU MSMs over the shards


@pytest.mark.gpu
def test_bench_one_process_per_gpu_leg_rehearsed_with_two_ranks():
    """`bench.py --gpus 2` end to end on the ONE GPU of the test box -- the self-launch under
    torch.distributed.run, two ranks (sharing GPU 0, collectives over gloo: ICICLE_BENCH_GLOO_REHEARSAL), the weak line, the
    strong object, the split transform with its all-to-alls, the watchdogs, and the in-process leg (virtual slots) -- so that
    every line of the N > 1 path has executed before the driver's 8-GPU node runs it over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ICICLE_BENCH_GLOO_REHEARSAL"] = "1"
    env["ICICLE_BENCH_INPROC_VIRTUAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size-log2", "18",
                        "--ntt-log2", "16", "--ntt-batch", "8"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert "error" not in out, out
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and "rehearsal" in out
    assert out["strong"]["value"] > 0 and out["strong"]["n_gpus"] == 2, out.get("strong")
    assert out["ntt"]["roundtrip_ok"] is True and out["ntt"]["value"] > 0
    assert out["ntt_split"]["roundtrip_ok"] is True, out.get("ntt_split")
    assert out["inproc"]["result_ok"] is True and out["inproc"]["n_gpus"] == 2, out.get("inproc")
