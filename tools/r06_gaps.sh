#!/bin/bash
# timeline of one small MSM (kernels, gaps): tools/r06_gaps.sh <logn>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/prof_g$L
  rocprofv3 --kernel-trace -d /tmp/prof_g$L -o msm -- python $R/tools/msm_one.py bn254 $L 1 > /tmp/prof_g$L.log 2>&1
  DB=$(find /tmp/prof_g$L -name '*.db' | head -1)
  echo "## 2^$L"; python $R/tools/rocpd_gaps.py "$DB"
done
