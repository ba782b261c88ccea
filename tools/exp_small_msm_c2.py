import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["perf_matrix.py", "none"]
import importlib.util
spec = importlib.util.spec_from_file_location("pm", "tools/perf_matrix.py")
pm = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(pm)
except SystemExit:
    pass
for curve in ("bn254", "bls12_381"):
    for logn in (9, 11, 13, 14, 15):
        pm.msm_case(curve, logn)
        for c in (8, 10, 11, 13, 15, 16):
            pm.msm_case(curve, logn, c=c)
