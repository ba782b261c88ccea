# kernel-level breakdown of one BN254 2^26 MSM: tools/gpu_prof_msm.sh <outdir>
O=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/$1
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof -o msm -- python $R/tools/msm_one.py bn254 26 > $O/prof.log 2>&1
DB=$(find $O/prof -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" > $O/msm_kernel_stats.txt
find $O/prof -name '*.db' -delete
head -40 $O/msm_kernel_stats.txt
