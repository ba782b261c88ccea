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
Host cores are PARTITIONED between the lanes and the foreground (the first measurement ran every lane on all 256 threads of the
box: the foreground's subprocess tests took 15 x longer and the suite gained nothing): each lane pins itself -- and thereby
the worker threads the reference spawns, which inherit the mask -- to its share, and the pytest process keeps the rest. The
reference CPU MSM scales poorly beyond a few dozen threads anyway (256 threads are 8 x faster than 8), so the shares cost
the background legs little. Below 32 cores nothing is pinned.
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

# share of the host cores per lane (fractions of os.cpu_count(); the foreground keeps the first quarter)
LANE_SHARE = {"fg": (0.0, 0.25), "msm_big": (0.25, 0.60), "msm": (0.60, 0.80), "ntt_a": (0.80, 0.90), "ntt_b": (0.90, 1.0)}


_ALL_CORES = None  # the cores this process could use before anything was pinned


def lane_cores(lane):
    """cores of `lane` (None: no pinning on this host)"""
    global _ALL_CORES
    if (os.cpu_count() or 1) < 32 or not hasattr(os, "sched_setaffinity"):
        return None
    if _ALL_CORES is None:
        try:
            _ALL_CORES = sorted(os.sched_getaffinity(0))
        except Exception:
            return None
    lo, hi = LANE_SHARE.get(lane, LANE_SHARE["msm"])
    n = len(_ALL_CORES)
    return _ALL_CORES[int(lo * n):max(int(lo * n) + 1, int(hi * n))] or None


def pin_to_lane(lane):
    """pin the CALLING thread (pid 0 = this thread on Linux) to the lane's cores; threads it spawns inherit the mask"""
    cores = lane_cores(lane)
    if cores:
        try:
            os.sched_setaffinity(0, cores)
        except OSError:
            pass
    return cores


class RefPool:
    def __init__(self):
        self._futures = {}
        self._lanes = {}
        self._dir = None
        self._fg_before = None
        try:
            self._fg_before = os.sched_getaffinity(0)
            lane_cores("fg")  # records the cores available to this process before anything is pinned (main thread, unpinned)
        except Exception:
            pass
        self._ntt = {}  # lane -> {"jobs": [...], "proc": Popen}
        self._started = {}
        self.timings = {}  # key -> seconds the reference call took (MSM lanes), for the session summary
        self.finished_at = {}  # key -> seconds after the pool was created (MSM lanes: when the call returned; NTT: when joined)
        self._t0 = time.time()

    # ---------------------------------------------------------------- MSM: in-process threads
    def _lane(self, name):
        if name not in self._lanes:
            self._lanes[name] = ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"ref-{name}")
        return self._lanes[name]

    def submit_msm(self, key, curve, scalars: np.ndarray, bases: np.ndarray, lane="msm", **kw):
        """reference msm() on host arrays (kept alive by the closure); result(key) -> projective_t[batch]"""
        from oracle import ref

        def run():
            pin_to_lane(lane)
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

    def pin_foreground(self):
        """the pytest process (and the subprocesses its tests spawn) keep the foreground share of the cores"""
        pin_to_lane("fg")

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
        if self._fg_before:
            try:
                os.sched_setaffinity(0, self._fg_before)
            except OSError:
                pass
