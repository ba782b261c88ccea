#!/usr/bin/env bash
# kernel timeline of the pipelined MSM: tools/gpu_trace_groups.sh <NG> [extra env...] -> gpurun_out/trace_ng<NG>/
NG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_ng$NG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
env ICICLE_HIP_MSM_GROUPS=$NG "$@" rocprofv3 --kernel-trace -d $O/prof -o msm -- python $R/tools/msm_one.py bn254 26 > $O/prof.log 2>&1
DB=$(find $O/prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_overlap.py "$DB" > $O/overlap.txt; python $R/tools/rocpd_list.py "$DB" > $O/list.txt 2>&1; fi
find $O/prof -name '*.db' -delete
cat $O/overlap.txt; tail -60 $O/list.txt
