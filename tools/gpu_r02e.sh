set -x
mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests/test_gpu_reference_suite.py tests/test_gpu_msm.py tests/test_gpu_msm_g2.py -m gpu -q -x --deselect "tests/test_gpu_msm.py::test_msm_full_size_split_property" --durations=8 > gpurun_out/r02e/pytest.txt 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r02e/pytest.txt
