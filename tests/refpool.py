"""TEST INFRASTRUCTURE: background pool for the reference-CPU legs of the full-size GPU parity tests.

The full-size tests (BASELINE configs[1]..[4] and the sizes next to them) compare the GPU result with the reference CPU
backend run on the FULL inputs. Rounds 1-5 ran those CPU legs inline: ~530 s of a 900 s suite with the GPU idle. Now every
such test declares its inputs and reference call as a JOB (`@pytest.mark.refjob("key")` + an entry in the module's
REF_JOBS dict); tests/conftest.py starts all selected jobs at session start, moves the tests that join them to the end of
the session, and the test body only waits for the future where its assertion needs the expected value. No compare is
dropped: the same reference call on the same bytes, just earlier and beside the GPU work.

Lanes:
  * MSM jobs run on two in-process threads (ctypes releases the GIL; the reference's cpu_msm keeps no state between calls):
    one lane for the five-minute BLS12-381 2^28 job, one for all BN254 jobs in submission order;
  * NTT jobs run in worker PROCESSES (tests/ref_ntt_worker.py): the reference keeps ONE twiddle domain per field and
    process, and the foreground tests of the same session init / release it at other sizes. Inputs and outputs travel
    as .npy files in a scratch directory (page cache), jobs of one lane run in submission order.
Host cores: the reference CPU MSM does NOT scale with its worker count -- every worker owns a full set of buckets that has to be
cleared and merged (cpu_msm.hpp:78-100, 365-417). Measured on the 256-thread GPU box (profiles/r06_ref_scaling.txt, BN254 2^24):
8 workers 13.7 s, 16: 8.0 s, 32: 6.6 s, 64: 7.8 s, 256 (the default): 8.5 s -- thirty-two workers are the fastest AND cost a tenth
of the core-seconds of the default; at 2^26 more workers still help (33 s on 256, 213 s on 16), at 2.5 x the core-seconds. So every
MSM job runs on its own thread with MSMConfig.ext "n_threads" sized by its term count (8 .. 48, msm_threads()), all of them at once; the NTT workers (no such knob: ntt_cpu.h uses hardware_concurrency) are pinned to 32 cores each, and
the foreground's own reference calls default to 32 workers (oracle/ref.py, ICICLE_REF_MSM_THREADS). The first attempt of round 6
ran every lane on all 256 threads (the foreground's subprocess tests took 15 x longer), the second gave each lane a fixed share
of the cores with 256 workers each (the 2^26 references took 200 s instead of 33): profiles/r06_notes.md section 1.
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

# cores of the NTT worker processes (fractions of the cores available to this process); everything else is not pinned
LANE_SHARE = {"ntt_a": (0.75, 0.875), "ntt_b": (0.875, 1.0)}
_ALL_CORES = None


def lane_cores(lane):
    """cores an NTT worker lane is pinned to (None: no pinning on this host / for this lane)"""
    global _ALL_CORES
    if lane not in LANE_SHARE or (os.cpu_count() or 1) < 64 or not hasattr(os, "sched_setaffinity"):
        return None
    if _ALL_CORES is None:
        try:
            _ALL_CORES = sorted(os.sched_getaffinity(0))
        except Exception:
            return None
    lo, hi = LANE_SHARE[lane]
    n = len(_ALL_CORES)
    return _ALL_CORES[int(lo * n):max(int(lo * n) + 1, int(hi * n))] or None


def msm_threads(n_terms):
    """worker count of the reference MSM for a job of `n_terms` terms per MSM (0 = the reference's default: small hosts). At 2^24 the
    reference is fastest with 32 workers, at 2^26 it still gains from more (33 s on 256, 213 s on 16): the big jobs get 40-48, so that
    all jobs together leave the foreground a few dozen cores."""
    if (os.cpu_count() or 1) < 64:
        return 0
    return 48 if n_terms >= (1 << 27) else 40 if n_terms >= (1 << 26) else 24 if n_terms >= (1 << 25) else 16 if n_terms >= (1 << 24) else 8


class RefPool:
    def __init__(self):
        self._futures = {}
        self._lanes = {}
        self._dir = None
        self._ntt = {}  # lane -> {"jobs": [...], "proc": Popen}
        self._started = {}
        self.timings = {}  # key -> seconds the reference call took (MSM lanes), for the session summary
        self.finished_at = {}  # key -> seconds after the pool was created (MSM lanes: when the call returned; NTT: when joined)
        self._t0 = time.time()

    # ---------------------------------------------------------------- MSM: in-process threads
    def _lane(self, name):
        if name not in self._lanes:  # (every MSM job has a thread of its own: the worker count bounds what it takes)
            self._lanes[name] = ThreadPoolExecutor(max_workers=16, thread_name_prefix=f"ref-{name}")
        return self._lanes[name]

    def submit_msm(self, key, curve, scalars: np.ndarray, bases: np.ndarray, lane="msm", **kw):
        """reference msm() on host arrays (kept alive by the closure); result(key) -> projective_t[batch]"""
        from oracle import ref

        kw.setdefault("n_threads", msm_threads(scalars.size // 8 // max(1, kw.get("batch", 1))))

        def run():
            t0 = time.time()
            out = ref.RefCurve(curve).msm(scalars, bases, **kw)
            self.timings[key] = time.time() - t0
            self.finished_at[key] = time.time() - self._t0
            return out

        assert key not in self._futures, key
        self._futures[key] = self._lane(lane).submit(run)

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
            cores = lane_cores(lane)
            if cores:
                env["ICICLE_REFPOOL_CORES"] = ",".join(str(c) for c in cores)
            st["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ref_ntt_worker.py"), d, lane], stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
            self._started[lane] = True

    # ---------------------------------------------------------------- join
    def has(self, key):
        return key in self._futures

    def result(self, key, timeout=1500):
        f = self._futures[key]
        if isinstance(f, Future):
            return f.result(timeout=timeout)
        _, lane, nout = f
        d, st = self.scratch(), self._ntt[lane]
        assert lane in self._started, f"lane {lane} was never started"
        done = os.path.join(d, f"done_{key}")
        t0 = time.time()
        while not os.path.exists(done):
            rc = st["proc"].poll()
            if rc is not None and not os.path.exists(done):
                raise RuntimeError(f"reference NTT worker (lane {lane}) exited with {rc} before job {key}:\n" + open(os.path.join(d, f"lane_{lane}.log")).read()[-2000:])
            if time.time() - t0 > timeout:
                raise TimeoutError(f"reference NTT job {key} not done after {timeout} s")
            time.sleep(0.05)
        outs = [np.load(os.path.join(d, f"out_{key}_{i}.npy"), mmap_mode="r") for i in range(nout)]
        try:
            self.timings[key] = float(open(done).read() or 0)
        except ValueError:
            pass
        return outs if nout > 1 else outs[0]

    def drop(self, key):
        """free the files of a finished NTT job"""
        if self._dir:
            for f in os.listdir(self._dir):
                if f.startswith((f"in_{key}.", f"out_{key}_")):
                    os.unlink(os.path.join(self._dir, f))

    def close(self):
        for st in self._ntt.values():
            p = st.get("proc")
            if p is not None and p.poll() is None:
                p.kill()
                p.wait()
        for ex in self._lanes.values():
            ex.shutdown(wait=False, cancel_futures=True)
        if self._dir:
            shutil.rmtree(self._dir, ignore_errors=True)
            self._dir = None
