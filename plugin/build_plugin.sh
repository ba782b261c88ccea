#!/usr/bin/env bash
# Builds the reference-runtime plugin DSOs (the drop-in form of this backend) against the reference
# headers and the reference libraries built by oracle/build_ref.sh. Needs /root/reference; outputs
# travel to the GPU box under plugin/lib/backend/hip/ (git-ignored; nothing the product ships lives under oracle/):
#   libicicle_backend_hip_device.so           "HIP" DeviceAPI            (RTLD_GLOBAL by name)
#   libicicle_backend_hip_curve_<c>.so        msm + msm_precompute_bases
#   libicicle_backend_hip_field_<f>.so        ntt family + extension ntt
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
R="${ICICLE_REFERENCE_DIR:-/root/reference}/icicle"
REF="$ROOT/oracle/_ref"
OUT="$HERE/lib/backend/hip"
rm -rf "$REF/backend" # (rounds 1-3 installed the plugin inside the oracle tree)
HIPLIB="$ROOT/icicle_amd/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
if [ ! -d "$R" ]; then echo "build_plugin: $R not present -- using prebuilt plugin if any" >&2; exit 0; fi
[ -f "$REF/libicicle_device.so" ] || "$ROOT/oracle/build_ref.sh"
mkdir -p "$OUT"
FLAGS="-std=c++17 -O2 -fPIC -shared -w -I$R/include"
CXX="${ORACLE_CXX:-/opt/rocm/lib/llvm/bin/clang++}"   # curve/field parts are plain C++ (no HIP headers)
RP="-Wl,-rpath,\$ORIGIN/../../../../oracle/_ref:\$ORIGIN/../../../../icicle_amd/lib" # the reference runtime that loads us + libicicle_hip.so
echo "[plugin] device"
# host-only C++ against the HIP runtime API (no kernels here): plain clang++, the HIP headers need the platform define
$CXX $FLAGS -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$HERE/hip_backend_device.cpp" -L"$REF" -licicle_device -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib $RP -o "$OUT/libicicle_backend_hip_device.so"
for spec in bn254:1 bls12_381:2 bls12_377:3; do
  c=${spec%%:*}; id=${spec##*:}
  echo "[plugin] curve $c"
  $CXX $FLAGS -DCURVE_ID=$id -DFIELD_ID=$id -DCURVE=$c -DFIELD=$c -DICICLE_FFI_PREFIX=$c -DMSM=ON -DNTT=ON -DECNTT=ON -DG2_ENABLED "$HERE/hip_backend_curve.cpp" \
    -L"$REF" -licicle_curve_$c -licicle_device -L"$HIPLIB" -licicle_hip $RP -o "$OUT/libicicle_backend_hip_curve_$c.so"
done
# Grumpkin: MSM only (icicle/cmake/features.cmake:19) -- no G2, no ECNTT, no NTT over its scalar field
echo "[plugin] curve grumpkin"
$CXX $FLAGS -DCURVE_ID=5 -DFIELD_ID=5 -DCURVE=grumpkin -DFIELD=grumpkin -DICICLE_FFI_PREFIX=grumpkin -DMSM=ON "$HERE/hip_backend_curve.cpp" \
  -L"$REF" -licicle_curve_grumpkin -licicle_device -L"$HIPLIB" -licicle_hip $RP -o "$OUT/libicicle_backend_hip_curve_grumpkin.so"
echo "[plugin] field grumpkin (scalar field of the curve, vector ops only)"
$CXX $FLAGS -DFIELD_ID=5 -DFIELD=grumpkin -DICICLE_FFI_PREFIX=grumpkin "$HERE/hip_backend_field.cpp" \
  -L"$REF" -licicle_field_grumpkin -licicle_device -L"$HIPLIB" -licicle_hip $RP -o "$OUT/libicicle_backend_hip_field_grumpkin.so"
for spec in bn254:1 bls12_381:2 bls12_377:3 stark252:1002; do   # 256-bit fields: Montgomery conversion + NTT over 32-byte elements
  f=${spec%%:*}; id=${spec##*:}
  echo "[plugin] field $f (scalar field of the curve)"
  $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$f -DICICLE_FFI_PREFIX=$f -DNTT=ON -DHIP_PLUGIN_SCALAR_FIELD_256 "$HERE/hip_backend_field.cpp" \
    -L"$REF" -licicle_field_$f -licicle_device -L"$HIPLIB" -licicle_hip $RP -o "$OUT/libicicle_backend_hip_field_$f.so"
done
echo "[plugin] field goldilocks"
$CXX $FLAGS -DFIELD_ID=1005 -DFIELD=goldilocks -DICICLE_FFI_PREFIX=goldilocks -DNTT=ON -DEXT_FIELD=ON -DHIP_PLUGIN_SCALAR_FIELD_64 "$HERE/hip_backend_field.cpp" \
  -L"$REF" -licicle_field_goldilocks -licicle_device -L"$HIPLIB" -licicle_hip $RP -o "$OUT/libicicle_backend_hip_field_goldilocks.so"
for spec in babybear:1001 koalabear:1004; do
  f=${spec%%:*}; id=${spec##*:}
  echo "[plugin] field $f"
  $CXX $FLAGS -DFIELD_ID=$id -DFIELD=$f -DICICLE_FFI_PREFIX=$f -DNTT=ON -DEXT_FIELD=ON "$HERE/hip_backend_field.cpp" \
    -L"$REF" -licicle_field_$f -licicle_device -L"$HIPLIB" -licicle_hip $RP -o "$OUT/libicicle_backend_hip_field_$f.so"
done
ls -la "$OUT"
