#!/usr/bin/env bash
# kernel stats of one MSM shape: tools/gpu_prof_msm_case.sh <tag> <curve> <logn> <batch> -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; curve=$2; logn=$3; batch=$4; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace -d /tmp/prof_$tag -o msm -- python $R/tools/msm_one.py $curve $logn $batch > /tmp/prof_$tag.log 2>&1
DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" | grep -v "k_generate\|at6native" > $R/gpurun_out/${tag}_kernel_stats.txt
head -22 $R/gpurun_out/${tag}_kernel_stats.txt
