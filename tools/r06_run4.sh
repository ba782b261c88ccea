set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ntt_refspace.py tests/test_gpu_ecntt.py tests/test_gpu_ntt_layouts.py -q --durations=5 2>&1 | tail -25 | cut -c1-2500 | tee $O/pytest_ntt_ecntt.txt
timeout 900 python -m pytest tests/test_gpu_multi_rehearsal.py tests/test_gpu_msm.py -q -k "peer or combine or refuses or precompute or mixed or table" 2>&1 | tail -15 | cut -c1-2500 | tee $O/pytest_msm_new.txt
timeout 300 python tools/perf_matrix.py ecntt 2>/dev/null | grep -v amdgpu.ids | tee $O/ecntt_glv.txt
for v in 1 0; do echo "ICICLE_HIP_NTT_PAD_LANES=$v"; ICICLE_HIP_NTT_PAD_LANES=$v timeout 300 python tools/perf_matrix.py layouts 2>/dev/null | grep -v amdgpu.ids; done | tee $O/layouts_pad_ab.txt
for v in 1 0 1 0; do echo "ICICLE_HIP_MSM_REDUCE_BALANCE=$v"; ICICLE_HIP_MSM_REDUCE_BALANCE=$v python bench.py --no-cpu-baseline --no-shard-extras --no-ntt 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],2), d['roofline']['phases_ms'], 'roof', round(d['roofline']['alu']['roof']/1e9,2))"; done | tee $O/reduce_balance_ab.txt
for w in 2 3 2 3; do echo "bls12_381 2^25 ICICLE_HIP_MSM_ACC_WAVES=$w"; ICICLE_HIP_MSM_ACC_WAVES=$w python - <<'PY'
import sys, os, time, ctypes
sys.path.insert(0, os.getcwd())
import torch
from icicle_amd import msm as M, runtime
from icicle_amd._lib import MSMConfig, lib, check
runtime.set_device(0); dev = torch.device("cuda", 0)
n = 1 << 25
b = torch.empty((n, 24), dtype=torch.int32, device=dev)
check(lib.bls12_381_hip_generate_affine_points(b.data_ptr(), n, 3, True, None))
g = torch.Generator(device=dev); g.manual_seed(1)
s = torch.randint(-(2**31), 2**31, (n, 8), dtype=torch.int32, device=dev, generator=g); s[:, 7] = torch.randint(0, 0x73EDA753, (n,), dtype=torch.int32, device=dev, generator=g)
r = torch.empty(36, dtype=torch.int32, device=dev)
c = MSMConfig.default(); c.is_async = True
f = lambda: M.msm("bls12_381", s.data_ptr(), b.data_ptr(), c, results=r.data_ptr(), msm_size=n)
f(); torch.cuda.synchronize()
lib.icicle_hip_enable_kernel_timing(True)
tot, cnt = ctypes.c_double(), ctypes.c_int()
lib.icicle_hip_kernel_timing(0, True, ctypes.byref(tot), ctypes.byref(cnt))
t0 = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3 * 1e3
lib.icicle_hip_kernel_timing(0, True, ctypes.byref(tot), ctypes.byref(cnt))
print(f"  whole {dt:.2f} ms, k_accumulate {tot.value / 3:.2f} ms")
PY
done 2>/dev/null | grep -v amdgpu | tee $O/bls_acc_waves_ab.txt
