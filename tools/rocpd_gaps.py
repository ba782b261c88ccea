#!/usr/bin/env python3
"""Per-kernel timeline of the LAST MSM in a rocprofv3 --kernel-trace rocpd database: start offset, duration and the idle
gap in front of every launch -- where a mid-size MSM spends its 3 ms. tools/rocpd_gaps.py results.db"""
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
    starts = [s for n, s, e in rows if "k_digits" in n]
    t0 = starts[-1]
    rows = [r for r in rows if r[1] >= t0]
    end = max(e for _, _, e in rows)
    busy = sum(e - s for _, s, e in rows)
    print(f"last MSM: {len(rows)} launches, span {(end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, gaps {(end - t0 - busy) / 1e3:.1f} us")
    prev = t0
    for n, s, e in rows:
        short = re.sub(r"^_ZN10icicle_hipL?\d+", "", n)[:44]
        print(f"  +{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  run {(e - s) / 1e3:8.1f}  {short}")
        prev = e


if __name__ == "__main__":
    main(sys.argv[1])
