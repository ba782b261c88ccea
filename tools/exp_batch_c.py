#!/usr/bin/env python3
"""c sweep for batches of small MSMs (the reference's published shape 2^12 x 2^10 and the criterion shapes)."""
import importlib.util
import os
import sys

sys.path.insert(0, os.getcwd())
sys.argv = ["perf_matrix.py", "none"]
spec = importlib.util.spec_from_file_location("pm", "tools/perf_matrix.py")
pm = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(pm)
except SystemExit:
    pass
for logn, batch, cs in ((12, 1024, (7, 8, 9, 10, 11, 12)), (13, 128, (8, 9, 10, 11, 12)), (16, 16, (10, 11, 12, 13, 15)), (17, 128, (11, 12, 13, 15, 16)), (20, 16, (13, 15, 16, 17))):
    pm.msm_case("bn254", logn, batch=batch)
    for c in cs:
        print(f"   batch {batch}: ", end="")
        pm.msm_case("bn254", logn, batch=batch, c=c)
