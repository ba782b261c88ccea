#!/bin/bash
# HBM traffic per kernel from PMC, the way MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# --pmc passes (they do not fit one), only --kernel-trace next to them. Prints KiB per dispatch for the dominant kernels.
# Calibration kernels with known byte counts in the same run: k_diag_gather (bench.py's in-run gather roof: 2^27 random
# 64-byte gathers = 8 GiB fetched per launch, the access pattern of k_accumulate's base fetch) and the NTT passes
# (4 GiB read + 4 GiB written per pass at 2^24 x 64).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  O=$R/gpurun_out/pmc_$C
  rm -rf $O; mkdir -p $O
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-shard-extras > $O/log.txt 2>&1
  F=$(find $O -name "*counter_collection.csv" | head -1)
  python - "$F" "$C" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != sys.argv[2]:
        continue
    n = r["Kernel_Name"].split("(")[0]
    agg[n].append(float(r["Counter_Value"]))
for n, v in agg.items():
    if any(k in n for k in ("k_accumulate", "k_ntt_fast", "k_diag_gather", "k_a_scatter", "k_b_scatter", "k_digits")):
        print(f"{sys.argv[2]:10s} {n[-110:]:110s} launches {len(v):3d}  avg {sum(v) / len(v):14.0f} KiB  max {max(v):14.0f}")
PY
  rm -rf $O
done
