#!/usr/bin/env bash
# round-3 GPU lease C: ECNTT variant bisect, host-operand pipeline timings, bench line check
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
for v in default ecold ecnoquad oldasm; do
  if [ $v = default ]; then unset ICICLE_HIP_LIB; else export ICICLE_HIP_LIB=$PWD/icicle_amd/lib_$v/libicicle_hip.so; fi
  echo "=== $v" >> $O/ecntt_bisect.txt
  timeout 300 python -m pytest tests/test_gpu_ecntt.py -q -x 2>&1 | tail -4 >> $O/ecntt_bisect.txt
  timeout 200 python tools/perf_matrix.py ecntt 2>&1 | grep ecntt >> $O/ecntt_bisect.txt
done
unset ICICLE_HIP_LIB
timeout 120 python -m pytest tests/test_gpu_golden.py -q -x 2>&1 | tail -3 > $O/golden.txt
timeout 400 python tools/perf_matrix.py host > $O/host.txt 2>&1; echo "host rc=$?" >> $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/ecntt_bisect.txt $O/golden.txt $O/host.txt $O/summary.txt; cat $O/bench.json
