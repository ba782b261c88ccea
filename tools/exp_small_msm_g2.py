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
for logn in (8, 10, 12, 14, 16):
    pm.msm_case("bn254", logn, g2=True)
    for c in (6, 8, 10, 11, 13, 15, 16):
        pm.msm_case("bn254", logn, g2=True, c=c)
