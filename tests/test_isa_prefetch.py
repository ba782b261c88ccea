"""ISA lint (CPU only): the software prefetches of the hot kernels must still be prefetches in the shipped code object.

Round 2 found `k_ntt_fast` issuing the next batch row's loads and, a few instructions later, `s_waitcnt vmcnt(6)` ..
`vmcnt(0)` in front of butterflies that read none of them: the waitcnt pass had merged a "pending" loop-entry edge with
the back edge conservatively, and with an in-order counter that waits for the loads just issued -- every row paid a full
memory latency for two rounds without any test noticing (profiles/r02_notes.md section 6). The fix is source-level
(first row waited for before the loop, copy pinned behind the stores); this test keeps a compiler or source change from
silently undoing it. It disassembles the gfx950 code objects embedded in libicicle_hip.so and checks, in the row loop of
the 2^8-point column pass and row pass:
  * no `s_waitcnt vmcnt(k)` with k below the number of loads in flight between the block of next-row loads and the
    loop's barrier (i.e. nothing waits for the prefetch before the first round has been computed);
  * the wait that releases the prefetched row comes after the last store and tolerates the stores in flight
    (`vmcnt(n)`, n >= number of stores).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "icicle_amd", "lib", "libicicle_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
COL = "_ZN10icicle_hip10k_ntt_fastINS_15babybear_paramsELi4ELi2ELb0ELb0ELb0ELb0ELb0ELb0ELb0ELi0EEEvPKjPjS3_S3_NS_8PassDescENS_9NttLaunchEj"
RUN = "_ZN10icicle_hip10k_ntt_fastINS_15babybear_paramsELi4ELi2ELb0ELb0ELb0ELb0ELb0ELb0ELb0ELi2EEEvPKjPjS3_S3_NS_8PassDescENS_9NttLaunchEj"  # RN run pass (round 5)
ROW = "_ZN10icicle_hip10k_ntt_fastINS_15babybear_paramsELi4ELi2ELb1ELb0ELb0ELb0ELb1ELb0ELb0ELi0EEEvPKjPjS3_S3_NS_8PassDescENS_9NttLaunchEj"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("library or llvm-objdump not available")
    d = tmp_path_factory.mktemp("isa")
    so = shutil.copy(LIB, d / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=d)
    objs = sorted(str(p) for p in d.iterdir() if "amdgcn" in p.name)
    assert objs, "no gfx950 code objects found in the library"
    return objs


def disassemble(objs, symbol):
    for o in objs:
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={symbol}", o], capture_output=True, text=True).stdout
        lines = [ln.split("//")[0].strip() for ln in out.splitlines() if "\t" in ln]
        lines = [ln for ln in lines if ln and not ln.endswith(":")]
        if len(lines) > 100:
            return lines
    pytest.fail(f"kernel {symbol} not found in the library's code objects")


def row_loop(lines):
    """Instructions of the innermost loop that contains the LDS exchange barrier and global stores: from the target of the
    last backward branch to that branch."""
    # the row loop is the last region of the kernel: [first global_load after the per-block twiddle setup .. final store]
    bar = max(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
    loads_before = [i for i, ln in enumerate(lines[:bar]) if ln.startswith("global_load")]
    # walk back from the barrier to the contiguous group of next-row loads (no barrier in between)
    first = loads_before[-1]
    while first - 1 in loads_before or any(lines[j].startswith("global_load") for j in range(max(0, first - 12), first)):
        first = max(j for j in range(max(0, first - 12), first) if lines[j].startswith("global_load"))
    stores = [i for i, ln in enumerate(lines) if ln.startswith("global_store") and i > bar]
    return first, bar, stores


@pytest.mark.parametrize("symbol,loads_per_row,stores_per_row", [(COL, 16, 16), (ROW, 4, 4)])
def test_ntt_next_row_prefetch_is_not_waited_for_early(code_objects, symbol, loads_per_row, stores_per_row):
    lines = disassemble(code_objects, symbol)
    first, bar, stores = row_loop(lines)
    loads = [i for i in range(first, bar) if lines[i].startswith("global_load")]
    assert len(loads) == loads_per_row, (len(loads), loads_per_row)
    assert len(stores) == stores_per_row, (len(stores), stores_per_row)
    # between the last prefetch load and the barrier nothing may wait for (part of) the prefetch
    for i in range(loads[-1] + 1, bar):
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", lines[i])
        assert not (m and int(m.group(1)) < loads_per_row), f"line {i}: '{lines[i]}' waits for the loads issued {i - loads[-1]} instructions earlier"
    # the wait that releases the prefetched row: after the last store, and it lets the stores stay in flight
    tail = [(i, int(m.group(1))) for i in range(stores[-1], min(len(lines), stores[-1] + 40)) for m in [re.match(r"s_waitcnt.*vmcnt\((\d+)\)", lines[i])] if m]
    assert tail, "no vmcnt wait after the last store of the row loop"
    assert tail[0][1] >= stores_per_row, f"'{lines[tail[0][0]]}' also waits for this row's stores"
    # and nothing between the barrier and the last store drains the counter
    for i in range(bar, stores[-1]):
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", lines[i])
        assert not (m and int(m.group(1)) == 0), f"line {i}: '{lines[i]}' drains the memory counter in the middle of a row"
