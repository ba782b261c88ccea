#!/bin/bash
# round 6, run 12: quad-cooperative bucket reduction for small / mid-size MSMs: parity + timing A/B (ICICLE_HIP_MSM_REDUCE_QUAD=0 = the one-lane kernels)
mkdir -p gpurun_out/r06l
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_sharded.py tests/test_gpu_reference_suite.py tests/test_gpu_rust_suite.py -q -x -k "not reference_tests_can_draw" 2>&1 | tail -6 > gpurun_out/r06l/msm_tests.txt
for Q in 0 1; do
  echo "## ICICLE_HIP_MSM_REDUCE_QUAD=$Q" >> gpurun_out/r06l/midsize.txt
  ICICLE_HIP_MSM_REDUCE_QUAD=$Q timeout 300 python tools/exp_msm_midsize.py 8 10 12 14 16 18 20 21 22 2>/dev/null | grep "^bn254" >> gpurun_out/r06l/midsize.txt
done
bash tools/gpu_prof_msm_case.sh r06l_msm16 bn254 16 1 > /dev/null; bash tools/gpu_prof_msm_case.sh r06l_msm20 bn254 20 1 > /dev/null
cat gpurun_out/r06l/msm_tests.txt gpurun_out/r06l/midsize.txt; head -8 gpurun_out/r06l_msm16_kernel_stats.txt; head -8 gpurun_out/r06l_msm20_kernel_stats.txt
