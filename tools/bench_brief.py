#!/usr/bin/env python3
"""Print a one-line summary of a bench.py JSON line read from stdin. Usage: bench.py ... | tail -1 | tools/bench_brief.py LABEL"""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read())
out = [label, f"{d['value']:.3f} {d['unit']}", f"{d['ms_per_step']:.2f} ms/step", f"dominant {d['roofline']['avg_launch_ms']:.2f} ms"]
if "ntt" in d:
    out += [f"| ntt {d['ntt']['value']:.0f}/s", f"{d['ntt']['roofline']['avg_launch_ms']:.2f} ms/dir", f"frac {d['ntt']['roofline']['frac']:.3f}"]
print(" ".join(out))
