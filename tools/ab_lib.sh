#!/bin/bash
# A/B build of libicicle_hip.so for same-box comparisons: tools/ab_lib.sh <name> "<extra hipcc flags>"
# -> icicle_amd/lib_<name>/libicicle_hip.so ; select at run time with ICICLE_HIP_LIB=<path> (icicle_amd/_lib.py)
set -e
cd "$(dirname "$0")/../icicle_amd/csrc"
make -j8 OUT=../lib_$1 EXTRA="$2" 2>&1 | grep -E "error|Error" || true
ls -la ../lib_$1/libicicle_hip.so
