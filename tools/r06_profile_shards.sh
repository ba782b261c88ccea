#!/bin/bash
# Kernel tables (rocprofv3 --kernel-trace) and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes as MI355X_MICROARCH.md
# prescribes) of the per-GPU shares of BASELINE configs[3] and [4]:
#   config3_shard  BLS12-381 G1 MSM of 2^25 terms  (2^28 over 8 GPUs)
#   config4_shard  KoalaBear NTT 2^22 x 128 rows    (1024 rows over 8 GPUs), forward + inverse
# Output: gpurun_out/r06_config{3,4}_shard_{kernel_stats,pmc}.txt (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() { # tag, command...
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- "$@" > /tmp/prof_$tag.log 2>&1
  DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" | grep -v "k_generate\|at6native\|distribution" > $O/r06_${tag}_kernel_stats.txt
  head -14 $O/r06_${tag}_kernel_stats.txt
  rm -rf /tmp/prof_$tag
}
pmc() { # tag, kernel name filter (regex alternatives), command...
  tag=$1; filt=$2; shift 2
  : > $O/r06_${tag}_pmc.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_${tag}_$C
    rm -rf $D; mkdir -p $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o t -- "$@" > $D/log.txt 2>&1
    F=$(find $D -name "*counter_collection.csv" | head -1)
    python - "$F" "$C" "$filt" >> $O/r06_${tag}_pmc.txt <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != sys.argv[2]:
        continue
    agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for n, v in sorted(agg.items()):
    if re.search(sys.argv[3], n):
        print(f"{sys.argv[2]:10s} {n[-100:]:100s} launches {len(v):3d}  avg {sum(v) / len(v) * 1024 / 1e9:9.3f} GB per launch (raw counter x 1 KiB)")
PY
    rm -rf $D
  done
  cat $O/r06_${tag}_pmc.txt
}
stats config3_shard python $R/tools/msm_one.py bls12_381 25
pmc config3_shard "k_accumulate|k_a_scatter|k_b_scatter|k_digits|k_reduce_wave" python $R/tools/msm_one.py bls12_381 25
NTT_ONE_FIELD=koalabear NTT_ONE_ROUNDTRIP=1 stats config4_shard python $R/tools/ntt_one.py 22 128 3
NTT_ONE_FIELD=koalabear NTT_ONE_ROUNDTRIP=1 pmc config4_shard "k_ntt_fast" python $R/tools/ntt_one.py 22 128 3
