"""c sweep of small single MSMs (device-resident result: the same kernels as the host-result path up to the combine), after the quad
bucket reduction changed what a bucket costs: exp_small_msm_c.py [curve]"""
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
for logn in (6, 8, 10, 12, 14, 15, 16, 17, 18):
    pm.msm_case(curve, logn)
    for c in range(max(4, logn - 5), min(19, logn + 3)):
        pm.msm_case(curve, logn, c=c)
