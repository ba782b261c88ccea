"""CPU: every reference-runtime plugin DSO (plugin/build_plugin.sh -> plugin/lib/backend/hip) resolves all of its symbols
against libicicle_hip.so and the reference libraries it is linked to -- a missing `icicle_hip_<prefix>_*` alias for a curve or
field would otherwise only surface at call time on the GPU box (lazy binding)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGINS = sorted(glob.glob(os.path.join(ROOT, "plugin", "lib", "backend", "hip", "libicicle_backend_hip_*.so")))


@pytest.mark.skipif(not PLUGINS, reason="plugin not built (needs /root/reference)")
def test_plugin_dsos_resolve_every_symbol():
    names = {os.path.basename(p) for p in PLUGINS}
    for want in ("device", "curve_bn254", "curve_bls12_381", "curve_bls12_377", "curve_grumpkin", "field_babybear", "field_koalabear",
                 "field_goldilocks", "field_bn254", "field_bls12_381", "field_bls12_377", "field_grumpkin", "field_stark252"):
        assert f"libicicle_backend_hip_{want}.so" in names, want
    for p in PLUGINS:
        r = subprocess.run(["ldd", "-r", p], capture_output=True, text=True)
        out = r.stdout + r.stderr
        assert "not found" not in out, (p, out)
        assert "undefined symbol" not in out, (p, [l for l in out.splitlines() if "undefined" in l][:5])
