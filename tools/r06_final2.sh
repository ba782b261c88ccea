#!/bin/bash
# last pass of the round: GPU suite, PMC traffic of the final sources, the bench line (run after pmc_json has re-stamped the PMC file: two calls)
mkdir -p gpurun_out/final2
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/final2/gpu_pytest.txt 2>&1; echo "pytest rc=$?" > gpurun_out/final2/summary.txt
bash tools/pmc_traffic.sh > gpurun_out/final2/pmc_traffic.txt 2>&1
tail -3 gpurun_out/final2/gpu_pytest.txt; cat gpurun_out/final2/summary.txt; grep -c SIZE gpurun_out/final2/pmc_traffic.txt
