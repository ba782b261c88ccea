"""c sweep of mid-size single MSMs (2^19 .. 2^24), asynchronous device-resident result: exp_mid_msm_c.py [curve]"""
import sys, os
sys.path.insert(0, os.getcwd())
curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
sys.argv = ["perf_matrix.py", "none"]
import importlib.util
spec = importlib.util.spec_from_file_location("pm", "tools/perf_matrix.py")
pm = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(pm)
except SystemExit:
    pass
for logn in (19, 20, 21, 22, 23, 24):
    pm.msm_case(curve, logn)
    for c in (15, 16, 17, 19, 20):
        pm.msm_case(curve, logn, c=c)
