set -x
mkdir -p gpurun_out/r02g
tools/ubench/msm_ubench_noasm alu > gpurun_out/r02g/alu_noasm.txt 2>&1
tools/ubench/msm_ubench alu > gpurun_out/r02g/alu_asm.txt 2>&1
cat gpurun_out/r02g/alu_noasm.txt gpurun_out/r02g/alu_asm.txt
python tools/exp_acc_time.py 26:20 24:20 22:20 > gpurun_out/r02g/acc_time.txt 2>&1
cat gpurun_out/r02g/acc_time.txt
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_g2.py tests/test_gpu_golden.py tests/test_gpu_ntt_scalar.py tests/test_gpu_ecntt.py tests/test_gpu_vecops.py -m gpu -q -x --deselect "tests/test_gpu_msm.py::test_msm_full_size_split_property" > gpurun_out/r02g/pytest.txt 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r02g/pytest.txt
