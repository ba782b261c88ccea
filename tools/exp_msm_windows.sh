#!/usr/bin/env bash
# uniform vs mixed window widths, same box: tools/exp_msm_windows.sh -> gpurun_out/msm_windows_ab.txt
out=gpurun_out/msm_windows_ab.txt
: > $out
run() { echo "== 2^$1 $2" >> $out; env $2 timeout 600 python tools/exp_msm_groups.py $1 2>&1 | tail -1 >> $out; }
for logn in 26 25 24; do
  run $logn ICICLE_HIP_MSM_WINDOWS=-1
  run $logn ICICLE_HIP_MSM_WINDOWS=0
done
run 26 ICICLE_HIP_MSM_WINDOWS=-1
run 26 ICICLE_HIP_MSM_WINDOWS=0
run 23 ICICLE_HIP_MSM_WINDOWS=-1
run 23 ICICLE_HIP_MSM_WINDOWS=14
run 22 ICICLE_HIP_MSM_WINDOWS=-1
run 22 ICICLE_HIP_MSM_WINDOWS=15
run 22 ICICLE_HIP_MSM_WINDOWS=14
EXP_CURVE=bls12_381 run 25 ICICLE_HIP_MSM_WINDOWS=-1
EXP_CURVE=bls12_381 run 25 ICICLE_HIP_MSM_WINDOWS=0
cat $out
