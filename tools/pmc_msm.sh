#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the MSM kernels of one BN254 2^26 MSM: tools/pmc_msm.sh [label]  (ICICLE_HIP_LIB selects the library)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  O=$R/gpurun_out/pmc_$C
  rm -rf $O; mkdir -p $O
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o t -- python $R/tools/msm_one.py bn254 26 > $O/log.txt 2>&1
  F=$(find $O -name "*counter_collection.csv" | head -1)
  python - "$F" "$C" "${1:-}" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != sys.argv[2]:
        continue
    agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for n, v in agg.items():
    if any(k in n for k in ("k_accumulate", "k_a_scatter", "k_b_scatter", "k_digits", "k_reduce_segments")):
        print(f"{sys.argv[3]:8s} {sys.argv[2]:10s} {n[-60:]:60s} launches {len(v):3d}  avg {sum(v) / len(v) * 1024 / 1e9:9.3f} GB")
PY
  rm -rf $O
done
