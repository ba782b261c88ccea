#!/bin/bash
# round 6, run 10: k_final / k_final_horner on the complete projective quad doublings: parity (MSM files that reach them) + small-MSM timing
mkdir -p gpurun_out/r06j
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_sharded.py -q -x -k "not reference_tests_can_draw" 2>&1 | tail -6 > gpurun_out/r06j/msm_tests.txt
timeout 300 python tools/exp_msm_midsize.py 12 16 20 22 > gpurun_out/r06j/midsize.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-shard-extras 2>gpurun_out/r06j/bench.err | tail -1 > gpurun_out/r06j/bench.json
cat gpurun_out/r06j/msm_tests.txt gpurun_out/r06j/midsize.txt; python tools/bench_brief.py kfinal < gpurun_out/r06j/bench.json
