#!/bin/bash
# round 6, run 13: small-MSM window rule (c = 8 / 15 / 16) on top of the quad reduction: parity + timing
mkdir -p gpurun_out/r06n
timeout 1800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_sharded.py tests/test_gpu_reference_suite.py tests/test_gpu_rust_suite.py tests/test_gpu_plugin.py tests/test_gpu_multi_rehearsal.py -q -x 2>&1 | tail -6 > gpurun_out/r06n/msm_tests.txt
timeout 300 python tools/exp_msm_midsize.py 6 8 10 12 13 14 15 16 18 2>/dev/null | grep "^bn254" > gpurun_out/r06n/midsize.txt
timeout 300 python tools/perf_matrix.py curves 2>/dev/null | grep "^msm" > gpurun_out/r06n/curves.txt
cat gpurun_out/r06n/msm_tests.txt gpurun_out/r06n/midsize.txt gpurun_out/r06n/curves.txt
