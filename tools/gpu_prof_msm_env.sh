#!/usr/bin/env bash
# kernel stats of BN254 2^LOGN MSMs under an environment: tools/gpu_prof_msm_env.sh <tag> <logn> [ENV=...] -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; logn=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace -d /tmp/prof_$tag -o msm -- python $R/tools/msm_one.py bn254 $logn > /tmp/prof_$tag.log 2>&1
DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" | grep -v "k_generate\|at6native" > $O/${tag}_kernel_stats.txt
head -24 $O/${tag}_kernel_stats.txt
