#!/usr/bin/env bash
out=gpurun_out/ntt_big_parts.txt; : > $out
for o in 0 1 2; do echo "== ICICLE_HIP_NTT_PARTS_ORDER=$o" >> $out; ICICLE_HIP_NTT_PARTS_ORDER=$o python tools/perf_matrix.py ntt 2>&1 | grep "2^25\|2^26\|2^27\|2^20\|2^22" >> $out; done
cat $out
