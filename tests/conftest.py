import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# What the host really grants (the cgroup quota, not the visible hardware threads: tests/refpool.py effective_cores) bounds the threads
# of every reference call of this session: the MSM's worker count (oracle/ref.py, ICICLE_REF_MSM_THREADS) and the Taskflow stand-in's
# pool (oracle/shim/taskflow, ICICLE_TASKFLOW_SHIM_MAX_THREADS). 256 threads on 16 cores of quota only burn the quota faster.
try:
    from tests.refpool import effective_cores as _effective_cores

    if _effective_cores() < (os.cpu_count() or 1):
        os.environ.setdefault("ICICLE_REF_MSM_THREADS", str(_effective_cores()))
        os.environ.setdefault("ICICLE_TASKFLOW_SHIM_MAX_THREADS", str(2 * _effective_cores()))
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refjob(*keys, order=0): the test joins background reference-CPU jobs of tests/refpool.py; it runs at the end of the session, lower order first")


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """Tests that join a background reference job run LAST (shortest reference leg first), so the GPU works through the rest
    of the suite while the host cores compute the expected values (tests/refpool.py)."""
    late = [it for it in items if it.get_closest_marker("refjob")]
    if late:
        rest = [it for it in items if not it.get_closest_marker("refjob")]
        # the subprocess tests of bench.py (two ranks + gloo + their own reference legs) are the most sensitive to the first minutes,
        # when every background job is in its parallel phase: they go behind the other foreground tests
        rest.sort(key=lambda it: 1 if it.module.__name__.endswith("test_bench_cli") else 0)
        late.sort(key=lambda it: it.get_closest_marker("refjob").kwargs.get("order", 0))
        items[:] = rest + late
    config._refjob_items = late


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    pool = getattr(config, "_refpool", None)
    if pool is not None and (pool.timings or getattr(pool, "setup_s", 0)):
        terminalreporter.write_line("[refpool] reference-CPU legs (s): " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(pool.timings.items()))
                                    + f"; setup {getattr(pool, 'setup_s', 0):.1f}; finished (s after session start): "
                                    + ", ".join(f"{k} {v:.0f}" for k, v in sorted(getattr(pool, 'finished_at', {}).items())))


@pytest.fixture(scope="session")
def hip():
    """The product library bound to GPU 0. GPU tests fail loudly (no CPU fallback) if it is absent."""
    # torch (used by a few GPU tests for device tensors) ships its own HIP runtime: initialise it BEFORE
    # libicicle_hip.so pulls in /opt/rocm's, exactly as bench.py does, so the process has one runtime
    try:
        import torch

        torch.cuda.is_available() and torch.cuda.init()
    except Exception:
        pass
    import icicle_amd
    from icicle_amd import runtime

    assert runtime.get_device_count() >= 1, "no HIP device visible"
    runtime.set_device(0)
    return icicle_amd


@pytest.fixture(scope="session")
def refpool(request):
    """The session's background pool of reference-CPU jobs: every job a SELECTED test names with @pytest.mark.refjob is started
    here (inputs generated on the GPU, downloaded once, handed to the reference on background lanes)."""
    from tests.refpool import RefPool

    pool = RefPool()
    items = getattr(request.config, "_refjob_items", [])
    wanted, registry = [], {}
    for it in items:
        registry.update(getattr(it.module, "REF_JOBS", {}))
        for k in it.get_closest_marker("refjob").args:
            if k not in wanted:
                wanted.append(k)
    if wanted:
        import torch

        hip_ = request.getfixturevalue("hip")
        dev = torch.device("cuda", 0)
        t0 = time.time()
        starters = []  # (a starter may serve several keys: it runs once)
        for k in sorted(wanted, key=lambda k: registry[k][0]):
            if registry[k][1] not in starters:
                starters.append(registry[k][1])
        for fn in starters:
            fn(pool, hip_, dev)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        pool.start()
        pool.setup_s = time.time() - t0
    request.config._refpool = pool
    yield pool
    pool.close()


@pytest.fixture(scope="session", autouse=True)
def _refpool_autostart(request):
    """start the background reference jobs before the first test of a session that will need them"""
    if getattr(request.config, "_refjob_items", []):
        request.getfixturevalue("refpool")
    yield
