set -x
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests/test_gpu_reference_suite.py tests/test_gpu_plugin.py tests/test_gpu_vecops.py -m gpu -q --durations=8 > gpurun_out/r02d/pytest.txt 2>&1
echo "pytest rc=$?"; tail -60 gpurun_out/r02d/pytest.txt
