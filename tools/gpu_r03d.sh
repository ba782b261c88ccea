#!/usr/bin/env bash
# round-3 GPU lease D: rehearsal incl. the split NTT, co-residency trace of the pipelined MSM schedule
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multi_rehearsal.py -q -x --durations=5 > $O/rehearsal.txt 2>&1; echo "rehearsal rc=$?" | tee -a $O/summary.txt
cd /tmp
for ng in 1 3; do
  rm -rf $O/prof$ng
  ICICLE_HIP_MSM_GROUPS=$ng timeout 300 rocprofv3 --kernel-trace -d $O/prof$ng -o msm -- python $R/tools/msm_one.py bn254 26 > $O/prof$ng.log 2>&1
  DB=$(find $O/prof$ng -name '*.db' | head -1)
  echo "=== ICICLE_HIP_MSM_GROUPS=$ng" >> $O/overlap.txt
  [ -n "$DB" ] && python $R/tools/rocpd_overlap.py "$DB" >> $O/overlap.txt 2>&1
  find $O/prof$ng -name '*.db' -delete
done
cd $R
tail -8 $O/rehearsal.txt; cat $O/overlap.txt
