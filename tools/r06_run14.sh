#!/bin/bash
# round 6, run 14: quad bucket reduction for BN254 G2: parity + A/B timing
mkdir -p gpurun_out/r06o
timeout 1500 python -m pytest tests/test_gpu_msm_g2.py tests/test_gpu_msm.py -q -x -k "g2 or G2 or combine or reference_tests_can_draw" 2>&1 | tail -5 > gpurun_out/r06o/g2_tests.txt
for Q in 0 1; do
  echo "## ICICLE_HIP_MSM_REDUCE_QUAD=$Q" >> gpurun_out/r06o/g2_perf.txt
  ICICLE_HIP_MSM_REDUCE_QUAD=$Q timeout 300 python tools/perf_matrix.py g2 2>/dev/null | grep "^msm" >> gpurun_out/r06o/g2_perf.txt
done
cat gpurun_out/r06o/g2_tests.txt gpurun_out/r06o/g2_perf.txt
