#!/bin/bash
# FETCH_SIZE / WRITE_SIZE and kernel times of the passes of big transforms (2^26 x 16, 2^27 x 8, 2^24 x 64 for comparison)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cs in "24 64" "26 16" "27 8"; do
  set -- $cs
  tag=ntt_$1_$2
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- python $R/tools/ntt_one.py $1 $2 3 > /tmp/prof_$tag.log 2>&1
  DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
  echo "== 2^$1 x $2: kernel times" | tee -a $O/ntt_big_pmc.txt
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" | grep k_ntt_fast | cut -c1-140 | tee -a $O/ntt_big_pmc.txt
  rm -rf /tmp/prof_$tag
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${tag}_$C; rm -rf $D; mkdir -p $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o t -- python $R/tools/ntt_one.py $1 $2 2 > $D/log.txt 2>&1
    F=$(find $D -name "*counter_collection.csv" | head -1)
    python - "$F" "$C" <<'PY' | tee -a $O/ntt_big_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == sys.argv[2] and "k_ntt_fast" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for n, v in sorted(agg.items()):
    print(f"{sys.argv[2]:10s} {n[-95:]:95s} launches {len(v):3d}  avg {sum(v) / len(v) * 1024 / 1e9:8.3f} GB")
PY
    rm -rf $D
  done
done
