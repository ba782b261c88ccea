#!/usr/bin/env python3
"""Co-residency report from a rocprofv3 --kernel-trace rocpd database: for every kernel name, how much of its run time
overlapped with a k_accumulate launch, and how long at least two kernels were in flight. Used for the pipelined MSM
schedule (ICICLE_HIP_MSM_GROUPS > 1): tools/rocpd_overlap.py results.db"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    kcols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
    rows = [(re.sub(r"\(.*", "", n or "?"), s, e) for n, s, e in db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id")]
    rows.sort(key=lambda r: r[1])
    acc = [(s, e) for n, s, e in rows if "k_accumulate" in n]
    if not acc:
        print("no k_accumulate launches in the trace")
        return
    # the last MSM of the run: everything from the last k_digits* launch on
    starts = [s for n, s, e in rows if "k_digits" in n]
    t0 = starts[-1] if starts else rows[0][1]
    rows = [r for r in rows if r[1] >= t0]
    acc = [(s, e) for n, s, e in rows if "k_accumulate" in n]
    span = max(e for _, _, e in rows) - t0
    print(f"last MSM of the trace: {len(rows)} launches, {span / 1e6:.3f} ms from the first digit kernel to the last kernel end, {len(acc)} k_accumulate launches")
    agg = {}
    for n, s, e in rows:
        if "k_accumulate" in n:
            continue
        ov = sum(max(0, min(e, ae) - max(s, as_)) for as_, ae in acc)
        a = agg.setdefault(n, [0, 0, 0])
        a[0] += 1
        a[1] += e - s
        a[2] += ov
    print(f"{'kernel':48s} {'calls':>5s} {'run ms':>9s} {'of it beside k_accumulate':>26s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:48]:48s} {a[0]:5d} {a[1] / 1e6:9.3f} {a[2] / 1e6:14.3f} ms ({100 * a[2] / max(1, a[1]):5.1f} %)")
    print(f"k_accumulate total {sum(e - s for s, e in acc) / 1e6:.3f} ms in {len(acc)} launches: " + ", ".join(f"{(e - s) / 1e6:.2f}" for s, e in acc))
    # time with >= 2 kernels in flight
    ev = sorted([(s, 1) for _, s, e in rows] + [(e, -1) for _, s, e in rows])
    depth, last, multi = 0, ev[0][0], 0
    for t, d in ev:
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    print(f"time with two or more kernels in flight: {multi / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
