#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel table:
calls, total/avg/min/max duration. Usage: tools/rocpd_stats.py results.db [> profiles/summary.txt]"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    kcols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
    q = f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"
    agg = {}
    for name, st, en in db.execute(q):
        name = re.sub(r"\(.*", "", name or "?")
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values()) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'%':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:70]:70s} {a[0]:6d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e6:10.3f} {a[2]/1e6:10.3f} {a[3]/1e6:10.3f} {100*a[1]/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
