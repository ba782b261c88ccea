#!/bin/bash
# round 6, run 9: ECNTT with complete projective quad doublings (ec_dbl_quad.hpp): parity + timing
mkdir -p gpurun_out/r06i
timeout 900 python -m pytest tests/test_gpu_ecntt.py -q -x 2>&1 | tail -5 > gpurun_out/r06i/ecntt_tests.txt
timeout 600 python tools/perf_matrix.py ecntt > gpurun_out/r06i/ecntt_perf.txt 2>&1
cat gpurun_out/r06i/ecntt_tests.txt gpurun_out/r06i/ecntt_perf.txt
