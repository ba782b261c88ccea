#!/usr/bin/env python3
"""List the kernel dispatches of a rocprofv3 rocpd SQLite database in launch order: start offset, duration, name.
Usage: tools/rocpd_list.py results.db [substring]"""
import re
import sqlite3
import sys


def main(path, filt=None):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    kcols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
    rows = list(db.execute(f"select d.start, d.end, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    t0 = rows[0][0] if rows else 0
    for st, en, name in rows:
        name = re.sub(r"\(.*", "", name or "?")
        if filt and filt not in name:
            continue
        print(f"{(st - t0) / 1e6:12.3f} ms  {(en - st) / 1e6:9.3f} ms  {name[:110]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
