#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *unmodified* reference CPU backend (the oracle for
# the MSM/NTT hot path) from the sources where they lie under /root/reference into
# oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
#
# No reference source is copied into this repo. The only stand-in is oracle/shim/taskflow
# (scheduling only, no arithmetic; Taskflow v3.8.0 is fetched from the network by the
# reference's CMake, backend/cpu/CMakeLists.txt:19-33, which is impossible here).
# Source lists mirror icicle/cmake/target_editor.cmake:4-12,36-49,60-68 and
# icicle/backend/cpu/CMakeLists.txt:43-81 with only the NTT / EXT_FIELD / MSM features on.
#
# Usage: oracle/build_ref.sh [device|bn254|bls12_381|bls12_377|grumpkin|babybear|koalabear|stark252|goldilocks ...]   (default: all)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
R="${ICICLE_REFERENCE_DIR:-/root/reference}/icicle"
OUT="$HERE/_ref"
SHIM="$HERE/shim"
CXX="${ORACLE_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
[ -x "$CXX" ] || CXX=g++
FLAGS="-std=c++17 -O3 -fPIC -shared -pthread -w -I$R/include -I$R/backend/cpu/include -I$SHIM"

if [ ! -d "$R" ]; then
  echo "build_ref: $R not present (GPU box?) -- using prebuilt oracle/_ref if any" >&2
  exit 0
fi
mkdir -p "$OUT"

build_device() {
  [ "$OUT/libicicle_device.so" -nt "$0" ] && return 0
  echo "[ref] libicicle_device.so"
  $CXX $FLAGS $R/src/device_api.cpp $R/src/runtime.cpp $R/src/config_extension.cpp \
    $R/backend/cpu/src/cpu_device_api.cpp -ldl -o "$OUT/libicicle_device.so"
}

# $1 = name, $2 = FIELD_ID, $3 = extra defs
build_field() {
  local name=$1 id=$2 extra=$3
  [ "$OUT/libicicle_field_$name.so" -nt "$0" ] && return 0
  echo "[ref] libicicle_field_$name.so"
  $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$name -DICICLE_FFI_PREFIX=$name -DNTT=ON $extra \
    $R/src/fields/ffi_extern.cpp $R/src/vec_ops.cpp $R/src/matrix_ops.cpp \
    $R/src/program/program_c_api.cpp $R/src/symbol/symbol_api.cpp \
    $R/src/ntt.cpp $R/src/polynomials/polynomials.cpp $R/src/polynomials/polynomials_c_api.cpp \
    $R/src/polynomials/polynomials_abstract_factory.cpp \
    $R/backend/cpu/src/field/cpu_vec_ops.cpp $R/backend/cpu/src/field/cpu_matrix_ops.cpp \
    $R/backend/cpu/src/field/cpu_ntt.cpp $R/backend/cpu/src/polynomials/cpu_polynomial_backend.cpp \
    -L"$OUT" -licicle_device -Wl,-rpath,'$ORIGIN' -o "$OUT/libicicle_field_$name.so"
}

# a scalar field the reference gives no NTT (Grumpkin, icicle/cmake/features.cmake:19): vec-ops only
build_field_no_ntt() {
  local name=$1 id=$2
  [ "$OUT/libicicle_field_$name.so" -nt "$0" ] && return 0
  echo "[ref] libicicle_field_$name.so (no NTT)"
  $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$name -DICICLE_FFI_PREFIX=$name \
    $R/src/fields/ffi_extern.cpp $R/src/vec_ops.cpp $R/src/matrix_ops.cpp \
    $R/src/program/program_c_api.cpp $R/src/symbol/symbol_api.cpp \
    $R/backend/cpu/src/field/cpu_vec_ops.cpp $R/backend/cpu/src/field/cpu_matrix_ops.cpp \
    -L"$OUT" -licicle_device -Wl,-rpath,'$ORIGIN' -o "$OUT/libicicle_field_$name.so"
}

# $1 = name, $2 = CURVE_ID (== FIELD_ID of its scalar field)
build_curve() {
  local name=$1 id=$2
  [ "$OUT/libicicle_curve_$name.so" -nt "$0" ] && return 0
  echo "[ref] libicicle_curve_$name.so"
  $CXX $FLAGS -DCURVE_ID=$id -DFIELD_ID=$id -DCURVE=$name -DFIELD=$name -DICICLE_FFI_PREFIX=$name -DMSM=ON -DNTT=ON -DECNTT=ON -DG2_ENABLED \
    $R/src/curves/ffi_extern.cpp $R/src/curves/montgomery_conversion.cpp $R/src/msm.cpp $R/src/ecntt.cpp \
    $R/backend/cpu/src/curve/cpu_mont_conversion.cpp $R/backend/cpu/src/curve/cpu_msm.cpp \
    $R/backend/cpu/src/curve/cpu_ecntt.cpp \
    -L"$OUT" -licicle_field_$name -licicle_device -Wl,-rpath,'$ORIGIN' -o "$OUT/libicicle_curve_$name.so"
}

# a curve with an MSM only (Grumpkin): no G2, no ECNTT
build_curve_msm_only() {
  local name=$1 id=$2
  [ "$OUT/libicicle_curve_$name.so" -nt "$0" ] && return 0
  echo "[ref] libicicle_curve_$name.so (MSM only)"
  $CXX $FLAGS -DCURVE_ID=$id -DFIELD_ID=$id -DCURVE=$name -DFIELD=$name -DICICLE_FFI_PREFIX=$name -DMSM=ON \
    $R/src/curves/ffi_extern.cpp $R/src/curves/montgomery_conversion.cpp $R/src/msm.cpp \
    $R/backend/cpu/src/curve/cpu_mont_conversion.cpp $R/backend/cpu/src/curve/cpu_msm.cpp \
    -L"$OUT" -licicle_field_$name -licicle_device -Wl,-rpath,'$ORIGIN' -o "$OUT/libicicle_curve_$name.so"
}

targets=("$@")
[ ${#targets[@]} -eq 0 ] && targets=(device bn254 bls12_381 bls12_377 grumpkin babybear koalabear stark252 goldilocks)
build_device
for t in "${targets[@]}"; do
  case $t in
    device) ;;
    bn254) build_field bn254 1 "" & ;;
    bls12_381) build_field bls12_381 2 "" & ;;
    bls12_377) build_field bls12_377 3 "" & ;;
    grumpkin) build_field_no_ntt grumpkin 5 & ;;
    stark252) build_field stark252 1002 "" & ;;
    goldilocks) build_field goldilocks 1005 "-DEXT_FIELD=ON" & ;;
    babybear) build_field babybear 1001 "-DEXT_FIELD=ON" & ;;
    koalabear) build_field koalabear 1004 "-DEXT_FIELD=ON" & ;;
    *) echo "unknown target $t" >&2; exit 1 ;;
  esac
done
wait
for t in "${targets[@]}"; do
  case $t in
    bn254) build_curve bn254 1 & ;;
    bls12_381) build_curve bls12_381 2 & ;;
    bls12_377) build_curve bls12_377 3 & ;;
    grumpkin) build_curve_msm_only grumpkin 5 & ;;
  esac
done
wait
ls -la "$OUT"
