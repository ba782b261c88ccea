#!/usr/bin/env bash
# kernel list of one BabyBear NTT case: tools/gpu_prof_ntt_env.sh <tag> <logn> <batch> [ENV=...] -> gpurun_out/<tag>_ntt_list.txt
tag=$1; logn=$2; batch=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace -d /tmp/prof_$tag -o ntt -- python $R/tools/ntt_one.py $logn $batch 3 > /tmp/prof_$tag.log 2>&1
DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_list.py "$DB" | grep -v "at6native\|gen_twiddles" | tail -14 > $R/gpurun_out/${tag}_ntt_list.txt
cat $R/gpurun_out/${tag}_ntt_list.txt
