set -x
mkdir -p gpurun_out/r02i
python tools/exp_acc_time.py 26:20 24:20 22:20 > gpurun_out/r02i/acc_time.txt 2>&1
cat gpurun_out/r02i/acc_time.txt
python tools/perf_matrix.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r02i/perf_matrix.txt; cat gpurun_out/r02i/perf_matrix.txt
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_g2.py tests/test_gpu_golden.py tests/test_gpu_msm_sharded.py -m gpu -q -x --deselect "tests/test_gpu_msm.py::test_msm_full_size_split_property" > gpurun_out/r02i/pytest.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r02i/pytest.txt
