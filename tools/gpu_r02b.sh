set -x
mkdir -p gpurun_out/r02b
python tools/exp_acc_time.py 26:20 24:20 22:20 > gpurun_out/r02b/acc_time.txt 2>&1
cat gpurun_out/r02b/acc_time.txt
timeout 1500 python -m pytest tests/test_gpu_msm_sharded.py tests/test_gpu_msm.py tests/test_gpu_msm_g2.py tests/test_gpu_golden.py tests/test_gpu_plugin.py -m gpu -q -x --deselect "tests/test_gpu_msm.py::test_msm_full_size_split_property" --durations=8 > gpurun_out/r02b/pytest.txt 2>&1
echo "pytest rc=$?"
tail -30 gpurun_out/r02b/pytest.txt
