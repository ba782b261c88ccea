"""GPU: the drop-in form. The reference's unmodified runtime + frontend libraries (oracle/_ref) load
this backend with icicle_load_backend() and run MSM / NTT on device "HIP" against their own "CPU"
device in one process. Runs in a subprocess so that the reference runtime owns the process-wide
icicle_* symbols exactly as it would in a user's application."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_runtime_loads_hip_backend_and_matches_cpu(hip):
    plug = os.path.join(ROOT, "plugin", "lib", "backend", "hip", "libicicle_backend_hip_device.so")
    if not os.path.exists(plug):
        pytest.skip("plugin not built (plugin/build_plugin.sh needs /root/reference)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "plugin_driver.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PLUGIN OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
