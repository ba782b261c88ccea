#!/usr/bin/env bash
# round-3 GPU lease B: rehearsal tests (gate fix), pipelined MSM schedule (parity + A/B), ECNTT, precompute sweep
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multi_rehearsal.py -q -x --durations=8 > $O/rehearsal.txt 2>&1; echo "rehearsal rc=$?" | tee -a $O/summary.txt
ICICLE_HIP_MSM_GROUPS=3 timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_sharded.py tests/test_gpu_golden.py tests/test_gpu_msm_g2.py -q -x > $O/groups_parity.txt 2>&1; echo "groups parity rc=$?" | tee -a $O/summary.txt
for ng in 1 2 3 4 6; do
  ICICLE_HIP_MSM_GROUPS=$ng timeout 200 python tools/perf_matrix.py groups >> $O/groups_ab.txt 2>&1
done
echo "groups ab rc=$?" | tee -a $O/summary.txt
timeout 400 python -m pytest tests/test_gpu_ecntt.py -q -x > $O/ecntt_test.txt 2>&1; echo "ecntt test rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/perf_matrix.py ecntt > $O/ecntt_perf.txt 2>&1; echo "ecntt perf rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/perf_matrix.py precompute > $O/precompute.txt 2>&1; echo "precompute rc=$?" | tee -a $O/summary.txt
cat $O/summary.txt; tail -15 $O/rehearsal.txt; tail -5 $O/groups_parity.txt; cat $O/groups_ab.txt; tail -3 $O/ecntt_test.txt; cat $O/ecntt_perf.txt gpurun_out/ecntt_timing.txt; cat $O/precompute.txt
