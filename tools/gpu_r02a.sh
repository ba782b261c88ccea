set -x
mkdir -p gpurun_out/r02a
nproc > gpurun_out/r02a/host.txt; free -g >> gpurun_out/r02a/host.txt
timeout 300 tools/ubench/msm_ubench all > gpurun_out/r02a/msm_ubench.txt 2>&1
echo "ubench rc=$?"
timeout 1500 python -m pytest tests/test_gpu_msm_sharded.py tests/test_gpu_ntt_fullsize.py tests/test_gpu_msm.py -m gpu -q -x -k "not c_abi and (sharded or projective_sum or logical or fullsize or full_size or config2 or config4 or fused)" --durations=10 > gpurun_out/r02a/pytest.txt 2>&1
echo "pytest rc=$?"
tail -25 gpurun_out/r02a/pytest.txt
cat gpurun_out/r02a/msm_ubench.txt | head -30
