"""TEST INFRASTRUCTURE: background pool for the reference-CPU legs of the full-size GPU parity tests.

The full-size tests (BASELINE configs[1]..[4] and the sizes next to them) compare the GPU result with the reference CPU
backend run on the FULL inputs. Rounds 1-5 ran those CPU legs inline: ~530 s of a 900 s suite with the GPU idle. Now every
such test declares its inputs and reference call as a JOB (`@pytest.mark.refjob("key")` + an entry in the module's
REF_JOBS dict); tests/conftest.py starts all selected jobs at session start, moves the tests that join them to the end of
the session, and the test body only waits for the future where its assertion needs the expected value. No compare is
dropped: the same reference call on the same bytes, just earlier and beside the GPU work.

Lanes:
  * every MSM job runs in a worker PROCESS of its own (tests/ref_msm_worker.py), started the moment it is submitted. Threads of the
    pytest process were tried first (ctypes releases the GIL): eight concurrent jobs then shared ONE address space, and the reference
    clears and merges gigabytes of per-worker bucket arrays -- every page fault of every job queued on the same mm lock. The box sat
    at 10 busy cores of 256 while a 2^26 job took 220 s (33 s alone) and a batch of 600 small MSMs 400 s (15 s alone), whatever the
    worker count (gpurun_out/r06e, profiles/r06_notes.md section 1);
  * NTT jobs run in worker processes too (tests/ref_ntt_worker.py), one per lane, jobs of a lane in submission order: the
    reference keeps ONE twiddle domain per field and process, and the foreground tests of the same session init / release it at
    other sizes.
  Inputs and outputs travel as .npy files in a scratch directory (a tmpfs on the GPU boxes), mapped by the workers.
Host cores: the GPU boxes of this build show 256 hardware threads but their cgroup grants 16 cores of CPU time (cpu.max = 1600000
100000, tools/cpu_probe.sh: 256 busy processes reach a parallelism of 15.7). Every reference leg of rounds 1-5 ran 256 threads on that
quota and burnt most of it on itself (every MSM worker owns, clears and merges a full bucket set; the NTT asks for
hardware_concurrency() threads per call). Five schedules that left the thread counts alone cost 721-829 s (profiles/r06_notes.md
section 1); with every reference call sized to the quota -- few workers per MSM job (msm_threads), the NTT workers' and the
foreground's calls capped (ICICLE_TASKFLOW_SHIM_MAX_THREADS in oracle/shim/taskflow, ICICLE_REF_MSM_THREADS in oracle/ref.py) -- the
same compares cost 478 s.
Nothing here is imported by the product.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time
from concurrent.futures import Future, ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def effective_cores():
    """CPU time this process can really get: the cgroup quota (cpu.max) if there is one, else the visible cores. The GPU boxes of this
    build show 256 hardware threads and grant 16 cores of quota (cat /sys/fs/cgroup/cpu.max -> 1600000 100000; tools/cpu_probe.sh)."""
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


def msm_threads(n_terms):
    """worker count of the reference MSM for a job of `n_terms` terms per MSM. All jobs run at once and share the cores the cgroup
    grants, so what matters is core-seconds per job, and the reference is most frugal with FEW workers (every worker owns, clears and
    merges a full bucket set, cpu_msm.hpp:78-100): profiles/r06_ref_scaling.txt, BN254 2^24: 8 workers 110 core-s, 16: 129, 256: 135
    at the quota. The one job that is the critical path on its own (2^28) gets as many workers as there are cores."""
    e = effective_cores()
    if e <= 8:
        return 0  # small hosts: the reference's default
    return e if n_terms >= (1 << 27) else max(4, e // 2) if n_terms >= (1 << 25) else max(2, e // 4)


def ntt_worker_threads():
    """thread cap of an NTT worker process (ICICLE_TASKFLOW_SHIM_MAX_THREADS: the reference asks for hardware_concurrency() threads)"""
    return max(2, effective_cores() // 2)


class RefPool:
    def __init__(self):
        self._futures = {}
        self._dir = None
        self._msm = {}  # key -> Popen
        self._ntt = {}  # lane -> {"jobs": [...], "proc": Popen}
        self._started = {}
        self.timings = {}  # key -> seconds the reference call took (MSM lanes), for the session summary
        self.finished_at = {}  # key -> seconds after the pool was created (MSM lanes: when the call returned; NTT: when joined)
        self._t0 = time.time()

    # ---------------------------------------------------------------- MSM: one worker process per job
    def submit_msm(self, key, curve, scalars: np.ndarray, bases: np.ndarray, lane="msm", **kw):
        """reference msm() on host arrays in a process of its own (tests/ref_msm_worker.py), started at once; result(key) ->
        projective_t[batch]. The arrays travel as .npy files in the scratch directory (a tmpfs on the GPU boxes) and are mapped,
        not copied, by the worker."""
        assert key not in self._futures, key
        d = self.scratch()
        np.save(os.path.join(d, f"in_{key}_scalars.npy"), np.ascontiguousarray(scalars))
        np.save(os.path.join(d, f"in_{key}_bases.npy"), np.ascontiguousarray(bases))
        kw.setdefault("n_threads", msm_threads(scalars.size // 8 // max(1, kw.get("batch", 1))))
        with open(os.path.join(d, f"msm_{key}.json"), "w") as f:
            json.dump({"key": key, "curve": curve, "kw": kw}, f)
        log = open(os.path.join(d, f"msm_{key}.log"), "w")
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ref_msm_worker.py"), d, key], stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
        self._msm[key] = proc
        self._futures[key] = ("msm", key, 1)

    # ---------------------------------------------------------------- NTT: worker processes
    def scratch(self):
        if self._dir is None:
            base = None
            try:  # a roomy /dev/shm (the GPU boxes: 1.5 TB) beats the container's overlay disk for multi-GiB .npy files
                st = os.statvfs("/dev/shm")
                if st.f_bavail * st.f_frsize > (64 << 30):
                    base = "/dev/shm"
            except OSError:
                pass
            self._dir = tempfile.mkdtemp(prefix="icicle_refpool_", dir=base)
        return self._dir

    def submit_ntt(self, key, field, x: np.ndarray, logn, direction, batch=1, ordering=0, coset_gen=1, lane="ntt", chain=None):
        """reference <field>_ntt on `x` (uint32, batch rows of 2^logn). `chain`: a list of further (direction, ordering, coset_gen)
        calls applied to the previous OUTPUT in the worker; result(key) then returns the list of all outputs."""
        assert key not in self._futures and lane not in self._started, (key, lane)
        d = self.scratch()
        np.save(os.path.join(d, f"in_{key}.npy"), np.ascontiguousarray(x))
        spec = {"key": key, "field": field, "logn": logn, "direction": direction, "batch": batch, "ordering": ordering,
                "coset_gen": coset_gen, "chain": chain or []}
        self._ntt.setdefault(lane, {"jobs": []})["jobs"].append(spec)
        self._futures[key] = ("ntt", lane, 1 + len(spec["chain"]))

    def start(self):
        """spawn the worker processes (after every submit_ntt of the session)"""
        for lane, st in self._ntt.items():
            if lane in self._started:
                continue
            d = self.scratch()
            with open(os.path.join(d, f"lane_{lane}.json"), "w") as f:
                json.dump(st["jobs"], f)
            log = open(os.path.join(d, f"lane_{lane}.log"), "w")
            env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            env["ICICLE_TASKFLOW_SHIM_MAX_THREADS"] = str(ntt_worker_threads())
            st["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ref_ntt_worker.py"), d, lane], stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
            self._started[lane] = True

    # ---------------------------------------------------------------- join
    def has(self, key):
        return key in self._futures

    def result(self, key, timeout=1500):
        f = self._futures[key]
        kind, lane, nout = f
        d = self.scratch()
        proc = self._msm[key] if kind == "msm" else self._ntt[lane]["proc"]
        logf = os.path.join(d, f"msm_{key}.log" if kind == "msm" else f"lane_{lane}.log")
        assert kind == "msm" or lane in self._started, f"lane {lane} was never started"
        done = os.path.join(d, f"done_{key}")
        t0 = time.time()
        while not os.path.exists(done):
            rc = proc.poll()
            if rc is not None and not os.path.exists(done):
                raise RuntimeError(f"reference worker for job {key} exited with {rc}:\n" + open(logf).read()[-2000:])
            if time.time() - t0 > timeout:
                raise TimeoutError(f"reference job {key} not done after {timeout} s")
            time.sleep(0.05)
        try:
            parts = open(done).read().split()
            self.timings[key] = float(parts[0])
            if len(parts) > 1:
                self.finished_at[key] = float(parts[1]) - self._t0
        except (ValueError, IndexError):
            pass
        if kind == "msm":
            return np.load(os.path.join(d, f"out_{key}.npy"))
        outs = [np.load(os.path.join(d, f"out_{key}_{i}.npy"), mmap_mode="r") for i in range(nout)]
        return outs if nout > 1 else outs[0]

    def drop(self, key):
        """free the files of a finished NTT job"""
        if self._dir:
            for f in os.listdir(self._dir):
                if f.startswith((f"in_{key}.", f"in_{key}_", f"out_{key}_", f"out_{key}.")):
                    os.unlink(os.path.join(self._dir, f))

    def close(self):
        for p in [st.get("proc") for st in self._ntt.values()] + list(self._msm.values()):
            if p is not None and p.poll() is None:
                p.kill()
                p.wait()
        if self._dir:
            shutil.rmtree(self._dir, ignore_errors=True)
            self._dir = None
