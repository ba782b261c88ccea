#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the gfx950 code object inside a hipcc object file or shared library
(read from the AMDGPU metadata note). `tools/kernel_regs.py icicle_amd/lib/ntt_lanes.o [substring]` prints one line per
kernel; kernels that spill (scratch > 0) are marked. CPU-only: needs the ROCm llvm tools."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path, tmp):
    """gfx950 ELF(s) embedded in `path` (a bundled .o, or a .so carrying a fat binary)"""
    so = os.path.join(tmp, "lib.so")
    subprocess.run(["cp", path, so], check=True)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def kernels(co):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    for ln in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            yield cur
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "agpr_count", "max_flat_workgroup_size"):
            cur[k] = v
    if cur.get("name"):
        yield cur


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        rows = [k for co in code_objects(path, tmp) for k in kernels(co)]
    dm = demangle([r["name"] for r in rows])
    for r in sorted(rows, key=lambda r: dm[r["name"]]):
        name = dm[r["name"]]
        if sub and sub not in name:
            continue
        name = re.sub(r"\(.*", "", name).replace("icicle_hip::", "").replace("void ", "")
        spill = int(r.get("private_segment_fixed_size", 0)) > 0
        print(f"{'SPILL ' if spill else '      '}vgpr {r.get('vgpr_count', '?'):>4} agpr {r.get('agpr_count', '?'):>3} sgpr {r.get('sgpr_count', '?'):>4} scratch {r.get('private_segment_fixed_size', '?'):>5} lds {r.get('group_segment_fixed_size', '?'):>6}  {name}")


if __name__ == "__main__":
    main()
