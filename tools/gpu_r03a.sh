#!/usr/bin/env bash
# round-3 GPU lease A: new rehearsal tests, full-size configs 3/4, strided ubench (64-B runs), CU-mask phase scaling, full suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multi_rehearsal.py -q -x > $O/rehearsal.txt 2>&1; echo "rehearsal rc=$?" | tee -a $O/summary.txt
timeout 120 tools/ubench/strided_ubench > $O/strided_ubench.txt 2>&1; echo "ubench rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/exp_cumask.py 26 > $O/cumask.txt 2>&1; echo "cumask rc=$?" | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize_configs.py -q -x --durations=5 > $O/fullsize.txt 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize_configs.py --deselect tests/test_gpu_multi_rehearsal.py --durations=15 > $O/suite.txt 2>&1; echo "suite rc=$?" | tee -a $O/summary.txt
tail -3 $O/rehearsal.txt $O/fullsize.txt $O/suite.txt; cat $O/cumask.txt; tail -32 $O/strided_ubench.txt
