#!/usr/bin/env bash
# window groups x co-resident sort A/B on one box: tools/exp_msm_groups.sh [LOGN] -> gpurun_out/msm_groups_ab.txt
logn=${1:-26}
out=gpurun_out/msm_groups_ab_$logn.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 600 python tools/exp_msm_groups.py $logn 2>&1 | tail -1 >> $out; }
run ICICLE_HIP_MSM_GROUPS=1
run ICICLE_HIP_MSM_GROUPS=3 ICICLE_HIP_MSM_CORESIDENT=0
for ng in 2 3 4 5 6 8; do run ICICLE_HIP_MSM_GROUPS=$ng; done
run ICICLE_HIP_MSM_GROUPS=1
cat $out
