#!/usr/bin/env bash
# A/B for VERDICT r04 item 1: the Rust suite's check_msm (sync copy BEFORE stream.synchronize) with rounds 1-4's
# hipStreamNonBlocking user streams against the blocking streams of round 5. Writes gpurun_out/stream_order_ab.txt
out=gpurun_out/stream_order_ab.txt
: > $out
for mode in 1 0; do
  for t in "check_msm bn254 20" "check_ntt_async_copy_before_sync babybear 20"; do
    echo "== ICICLE_HIP_STREAMS_NONBLOCKING=$mode  $t" >> $out
    ICICLE_HIP_STREAMS_NONBLOCKING=$mode timeout 600 python tests/rust_suite_driver.py $t 2>&1 | grep -v DEBUG | tail -3 >> $out
  done
done
cat $out
