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
for c in (8, 10, 12, 13, 14, 15, 16, 17, 18, 20):
    pm.msm_case("bn254", 10, c=c)
