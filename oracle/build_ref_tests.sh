#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- compiles the reference's OWN test sources, unmodified and from where they lie under
# /root/reference/icicle/tests, against the reference libraries of oracle/build_ref.sh and the GoogleTest stand-in
# oracle/shim/gtest (GoogleTest itself is fetched from the network by the reference's CMake). Outputs travel to the
# GPU box under oracle/_ref/tests/ (git-ignored):
#   test_device_api                      icicle/tests/test_device_api.cpp
#   test_curve_api_{bn254,bls12_381,bls12_377}  icicle/tests/test_curve_api.cpp  (-DMSM -DG2_ENABLED -DECNTT, no PAIRING)
#   test_curve_api_grumpkin              the same source with -DMSM only
#   test_modarith_{babybear,koalabear,goldilocks,bn254,bls12_381,bls12_377,stark252}  icicle/tests/test_mod_arithmetic_api.h via oracle/shim/tests/modarith_main.cpp
#   example_msm, example_ntt             examples/c++/{msm,ntt}/example.cpp (bn254), run as `example_msm HIP`
#   example_best_practice_ntt            examples/c++/best-practice-ntt/example.cpp (three streams)
# Run with ICICLE_BACKEND_INSTALL_DIR=plugin/lib/backend so that the reference runtime loads the HIP plugin and makes
# "HIP" the main device (icicle/tests/test_base.h:37-46): tests/test_gpu_reference_suite.py.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
R="${ICICLE_REFERENCE_DIR:-/root/reference}/icicle"
REF="$HERE/_ref"
OUT="$REF/tests"
if [ ! -d "$R" ]; then echo "build_ref_tests: $R not present -- using prebuilt tests if any" >&2; exit 0; fi
[ -f "$REF/libicicle_device.so" ] || "$HERE/build_ref.sh"
mkdir -p "$OUT"
CXX="${ORACLE_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
[ -x "$CXX" ] || CXX=g++
FLAGS="-std=c++17 -O2 -pthread -w -I$HERE/shim/gtest -I$R/include -I$R/tests -I$R/backend/cpu/include -I$HERE/shim"
RP="-Wl,-rpath,\$ORIGIN/.."
newer() { [ "$1" -nt "$0" ] && [ "$1" -nt "$HERE/shim/gtest/gtest/gtest.h" ]; }

newer "$OUT/test_device_api" || { echo "[ref-tests] test_device_api"; $CXX $FLAGS "$R/tests/test_device_api.cpp" -L"$REF" -licicle_device $RP -o "$OUT/test_device_api" & }
for spec in bn254:1 bls12_381:2 bls12_377:3; do
  c=${spec%%:*}; id=${spec##*:}
  newer "$OUT/test_curve_api_$c" || { echo "[ref-tests] test_curve_api_$c"
    $CXX $FLAGS -DCURVE_ID=$id -DFIELD_ID=$id -DCURVE=$c -DFIELD=$c -DICICLE_FFI_PREFIX=$c -DMSM=ON -DNTT=ON -DECNTT=ON -DG2_ENABLED \
      "$R/tests/test_curve_api.cpp" -L"$REF" -licicle_curve_$c -licicle_field_$c -licicle_device $RP -o "$OUT/test_curve_api_$c" & }
  newer "$OUT/test_modarith_$c" || { echo "[ref-tests] test_modarith_$c"
    $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$c -DICICLE_FFI_PREFIX=$c -DNTT=ON \
      "$HERE/shim/tests/modarith_main.cpp" -L"$REF" -licicle_field_$c -licicle_device $RP -o "$OUT/test_modarith_$c" & }
done
# Grumpkin: an MSM and nothing else (icicle/cmake/features.cmake:19); stark252: a 252-bit field with an NTT
newer "$OUT/test_curve_api_grumpkin" || { echo "[ref-tests] test_curve_api_grumpkin"
  $CXX $FLAGS -DCURVE_ID=5 -DFIELD_ID=5 -DCURVE=grumpkin -DFIELD=grumpkin -DICICLE_FFI_PREFIX=grumpkin -DMSM=ON \
    "$R/tests/test_curve_api.cpp" -L"$REF" -licicle_curve_grumpkin -licicle_field_grumpkin -licicle_device $RP -o "$OUT/test_curve_api_grumpkin" & }
newer "$OUT/test_modarith_stark252" || { echo "[ref-tests] test_modarith_stark252"
  $CXX $FLAGS -DFIELD_ID=1002 -DFIELD=stark252 -DICICLE_FFI_PREFIX=stark252 -DNTT=ON \
    "$HERE/shim/tests/modarith_main.cpp" -L"$REF" -licicle_field_stark252 -licicle_device $RP -o "$OUT/test_modarith_stark252" & }
for spec in babybear:1001 koalabear:1004 goldilocks:1005; do
  f=${spec%%:*}; id=${spec##*:}
  newer "$OUT/test_modarith_$f" || { echo "[ref-tests] test_modarith_$f"
    $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$f -DICICLE_FFI_PREFIX=$f -DNTT=ON -DEXT_FIELD=ON \
      "$HERE/shim/tests/modarith_main.cpp" -L"$REF" -licicle_field_$f -licicle_device $RP -o "$OUT/test_modarith_$f" & }
done
# MatrixTest.matrixTranspose (tests/test_matrix_api.h:450-505) for one field of every element width
for spec in babybear:1001:-DEXT_FIELD=ON koalabear:1004:-DEXT_FIELD=ON goldilocks:1005:-DEXT_FIELD=ON bn254:1:; do
  f=${spec%%:*}; rest=${spec#*:}; id=${rest%%:*}; extra=${rest#*:}
  newer "$OUT/test_matrix_$f" || { echo "[ref-tests] test_matrix_$f"
    $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$f -DICICLE_FFI_PREFIX=$f -DNTT=ON $extra \
      "$HERE/shim/tests/matrix_main.cpp" -L"$REF" -licicle_field_$f -licicle_device $RP -o "$OUT/test_matrix_$f" & }
done
# the reference's C++ examples for this path, unmodified (examples/c++/msm/example.cpp, examples/c++/ntt/example.cpp):
# `example_msm HIP` / `example_ntt HIP` select the device by name exactly as a user of the reference would
EX="${ICICLE_REFERENCE_DIR:-/root/reference}/examples/c++"
for e in msm ntt; do
  newer "$OUT/example_$e" || { echo "[ref-tests] example_$e"
    $CXX -std=c++17 -O2 -pthread -w -I$R/include -I$EX -DCURVE_ID=1 -DFIELD_ID=1 -DG2_ENABLED "$EX/$e/example.cpp" \
      -L"$REF" -licicle_curve_bn254 -licicle_field_bn254 -licicle_device $RP -o "$OUT/example_$e" & }
done
# examples/c++/best-practice-ntt/example.cpp: the three-stream upload / compute / download pattern (bn254 scalar field, 2^20 x 16)
newer "$OUT/example_best_practice_ntt" || { echo "[ref-tests] example_best_practice_ntt"
  $CXX -std=c++17 -O2 -pthread -w -I$R/include -I$EX -DCURVE_ID=1 -DFIELD_ID=1 "$EX/best-practice-ntt/example.cpp" \
    -L"$REF" -licicle_curve_bn254 -licicle_field_bn254 -licicle_device $RP -o "$OUT/example_best_practice_ntt" & }
wait
ls -la "$OUT"
