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


# ---- scratch lint: hot and warm kernels must not spill ---------------------------------------------------
# A kernel that keeps private arrays or spills registers pays twice on gfx950: the traffic itself, and the wave limit the scratch
# reservation imposes (k_precompute held 4.7 - 14.5 KB per lane through rounds 3-5 and ran at 211 ms against a 135 ms floor).
# (regex on the demangled kernel name, bytes of scratch per lane allowed)
SCRATCH_RULES = [
    (r"^k_accumulate<(bn254|grumpkin)_g1, 3, false>", 0),          # the dominant MSM kernel as launched (3 waves per SIMD)
    (r"^k_accumulate<(bls12_381|bls12_377)_g1, 2, false>", 0),     # 14-limb fields: launched at 2 waves per SIMD
    (r"^k_accumulate<bn254_g2, 2, false>", 128),                   # (the default G2 launch: 100 B today)
    (r"^k_precompute_(chains|affine)<\w+_g1, \d>", 0),
    (r"^k_precompute_(chains|affine)<\w+_g2, \d>", 256),
    (r"^k_(reduce_wave|reduce_small|reduce_window|final|final_horner|final_combine|fold_overflow|proj_sum|bucket_add|generate)<\w+_g1>", 0),
    (r"^k_(digits|digits_count|a_scatter|b_scatter|a_count|b_count|bsize_count|bsize_scatter|plan_overflow|tables_from_a|b_plan|scan_sums|scan_apply)\b", 0),
    (r"^k_diag_(madd|ntt_pass)<", 0),                              # the in-run roofs must not be measured on a spilling kernel
    # NTT passes of the headline shapes: plain / inverse two-round column and row passes, 4- and 16-byte lanes, lane-native tiles
    (r"^k_ntt_fast<\w+_params, 4, 2, (false|true), (false|true), false, false, (false|true), false, (false|true), 0>", 0),
    (r"^k_ntt_fast<\w+_params, \d, \d, false, false, false, false, false, false, (false|true), [12]>", 0),  # native bit-reversed input
]


def test_listed_kernels_do_not_use_scratch(tmp_path):
    import importlib.util
    import sys

    if not os.path.exists(LIB):
        pytest.skip("library not built")
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(ROOT, "tools", "kernel_regs.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = [k for co in kr.code_objects(LIB, str(tmp_path)) for k in kr.kernels(co)]
    dm = kr.demangle([r["name"] for r in rows])
    seen = {i: 0 for i in range(len(SCRATCH_RULES))}
    bad = []
    for r in rows:
        name = re.sub(r"\(.*", "", dm[r["name"]]).replace("icicle_hip::", "").replace("void ", "")
        for i, (pat, limit) in enumerate(SCRATCH_RULES):
            if re.search(pat, name):
                seen[i] += 1
                sc = int(r.get("private_segment_fixed_size", 0))
                if sc > limit:
                    bad.append(f"{name}: {sc} B of scratch per lane (allowed {limit}), {r.get('vgpr_count')} VGPRs")
    assert not bad, "\n".join(bad)
    unmatched = [SCRATCH_RULES[i][0] for i, c in seen.items() if c == 0]
    assert not unmatched, f"rules that match no kernel of the library (renamed?): {unmatched}"
