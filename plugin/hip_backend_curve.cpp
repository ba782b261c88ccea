// Reference-runtime plugin, part 2/3: MSM for one curve (compile once per curve with the reference's
// own defines: -DCURVE_ID=<n> -DFIELD_ID=<n> -DICICLE_FFI_PREFIX=<curve>, icicle/cmake/curve.cmake:43-76).
// Registers functions with exactly the MsmImpl / MsmPreComputeImpl signatures
// (icicle/include/icicle/backend/msm_backend.h:11-44) under device type "HIP"; each forwards to the
// collision-free C entry points of libicicle_hip.so (include/icicle_hip.h). MSMConfig is passed
// through byte-for-byte (same 40-byte layout) except `ext`: the reference's ConfigExtension object is
// translated key by key (HipExt, hip_c_api.h).
#include <cstring>
#include "icicle/backend/msm_backend.h"
#ifdef ECNTT
  #include "icicle/backend/ecntt_backend.h"
#endif
#include "icicle/curves/montgomery_conversion.h"
#include "icicle/curves/curve_config.h"
#include "icicle/utils/utils.h"
#include "hip_c_api.h"

using namespace curve_config;
using namespace icicle;

#define HIP_FN(name) CONCAT_EXPAND(CONCAT_EXPAND(icicle_hip, ICICLE_FFI_PREFIX), name)

static_assert(sizeof(MSMConfig) == sizeof(hip_msm_config_t), "MSMConfig layout drifted");

static hip_msm_config_t translate(const MSMConfig& c)
{
  hip_msm_config_t o;
  std::memcpy(&o, &c, sizeof(o));
  o.ext = nullptr; // foreign ConfigExtension object: tolerated and ignored
  return o;
}

static eIcicleError hip_msm(const Device& device, const scalar_t* scalars, const affine_t* bases, int msm_size, const MSMConfig& config, projective_t* results)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_msm_config_t c = translate(config);
  HipExt ext(config.ext); // "hip_num_devices" / "hip_msm_exchange_buckets" reach the backend, foreign keys do not
  c.ext = ext.h;
  return (eIcicleError)HIP_FN(msm)(scalars, bases, msm_size, &c, results);
}

static eIcicleError hip_msm_precompute(const Device& device, const affine_t* input_bases, int nof_bases, const MSMConfig& config, affine_t* output_bases)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_msm_config_t c = translate(config);
  return (eIcicleError)HIP_FN(msm_precompute_bases)(input_bases, nof_bases, &c, output_bases);
}

static_assert(sizeof(VecOpsConfig) == sizeof(hip_vec_ops_config_t), "VecOpsConfig layout drifted");
static hip_vec_ops_config_t translate(const VecOpsConfig& c)
{
  hip_vec_ops_config_t o;
  std::memcpy(&o, &c, sizeof(o));
  o.ext = nullptr;
  return o;
}
// affine / projective Montgomery conversion (icicle/include/icicle/curves/montgomery_conversion.h:21-52)
static eIcicleError hip_affine_convert(const Device& device, const affine_t* input, size_t n, bool is_into, const VecOpsConfig& config, affine_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_vec_ops_config_t c = translate(config);
  return (eIcicleError)HIP_FN(affine_convert_montgomery)(input, n, is_into, &c, output);
}
static eIcicleError hip_projective_convert(const Device& device, const projective_t* input, size_t n, bool is_into, const VecOpsConfig& config, projective_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_vec_ops_config_t c = translate(config);
  return (eIcicleError)HIP_FN(projective_convert_montgomery)(input, n, is_into, &c, output);
}
REGISTER_AFFINE_CONVERT_MONTGOMERY_BACKEND("HIP", hip_affine_convert);
REGISTER_PROJECTIVE_CONVERT_MONTGOMERY_BACKEND("HIP", hip_projective_convert);

REGISTER_MSM_BACKEND("HIP", hip_msm);
REGISTER_MSM_PRE_COMPUTE_BASES_BACKEND("HIP", hip_msm_precompute);

#ifdef G2_ENABLED
// G2 (icicle/include/icicle/backend/msm_backend.h:46-81, curves/montgomery_conversion.h:53-89)
#define HIP_G2_FN(name) CONCAT_EXPAND(CONCAT_EXPAND(CONCAT_EXPAND(icicle_hip, ICICLE_FFI_PREFIX), g2), name)

static eIcicleError hip_g2_msm(const Device& device, const scalar_t* scalars, const g2_affine_t* bases, int msm_size, const MSMConfig& config, g2_projective_t* results)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_msm_config_t c = translate(config);
  HipExt ext(config.ext);
  c.ext = ext.h;
  return (eIcicleError)HIP_G2_FN(msm)(scalars, bases, msm_size, &c, results);
}
static eIcicleError hip_g2_msm_precompute(const Device& device, const g2_affine_t* input_bases, int nof_bases, const MSMConfig& config, g2_affine_t* output_bases)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_msm_config_t c = translate(config);
  return (eIcicleError)HIP_G2_FN(msm_precompute_bases)(input_bases, nof_bases, &c, output_bases);
}
static eIcicleError hip_g2_affine_convert(const Device& device, const g2_affine_t* input, size_t n, bool is_into, const VecOpsConfig& config, g2_affine_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_vec_ops_config_t c = translate(config);
  return (eIcicleError)HIP_G2_FN(affine_convert_montgomery)(input, n, is_into, &c, output);
}
static eIcicleError hip_g2_projective_convert(const Device& device, const g2_projective_t* input, size_t n, bool is_into, const VecOpsConfig& config, g2_projective_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_vec_ops_config_t c = translate(config);
  return (eIcicleError)HIP_G2_FN(projective_convert_montgomery)(input, n, is_into, &c, output);
}
REGISTER_MSM_G2_BACKEND("HIP", hip_g2_msm);
REGISTER_MSM_G2_PRE_COMPUTE_BASES_BACKEND("HIP", hip_g2_msm_precompute);
REGISTER_AFFINE_G2_CONVERT_MONTGOMERY_BACKEND("HIP", hip_g2_affine_convert);
REGISTER_PROJECTIVE_G2_CONVERT_MONTGOMERY_BACKEND("HIP", hip_g2_projective_convert);
#endif // G2_ENABLED

#ifdef ECNTT
// ECNTT (icicle/include/icicle/backend/ecntt_backend.h:16-33): NTTConfig<scalar_t> is passed through byte-for-byte
static_assert(sizeof(NTTConfig<scalar_t>) == sizeof(hip_ntt_config_u256_t), "NTTConfig<scalar_t> layout drifted");
static eIcicleError hip_ecntt(const Device& device, const projective_t* input, int size, NTTDir dir, const NTTConfig<scalar_t>& config, projective_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_ntt_config_u256_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(ecntt)(input, size, (int)dir, &c, output);
}
REGISTER_ECNTT_BACKEND("HIP", hip_ecntt);
#endif // ECNTT
