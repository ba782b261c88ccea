// Reference-runtime plugin, part 3/3: per-field APIs: scalar_convert_montgomery and the NTT family.
// Compile once per field with -DFIELD_ID=<n> -DICICLE_FFI_PREFIX=<field> -DNTT=ON (icicle/cmake/field.cmake:42-79):
//   31-bit fields (babybear, koalabear): add -DEXT_FIELD=ON, the extension-field NTT is registered too;
//   256-bit fields (the scalar fields of bn254, bls12_381, bls12_377; stark252): add -DHIP_PLUGIN_SCALAR_FIELD_256;
//   goldilocks (8-byte elements, quadratic extension): add -DEXT_FIELD=ON -DHIP_PLUGIN_SCALAR_FIELD_64;
//   a field the reference gives no NTT (grumpkin's scalar field): leave -DNTT out, only the vector ops are registered.
// Registers all four members of the NTT API family (a missing member makes the reference dispatcher
// THROW through its extern "C" shim) plus the extension-field NTT, with the
// signatures of icicle/include/icicle/backend/ntt_backend.h:13-93.
#include <cstring>
#include "icicle/backend/ntt_backend.h"
#include "icicle/backend/vec_ops_backend.h"
#include "icicle/fields/field_config.h"
#include "icicle/utils/utils.h"
#include "hip_c_api.h"

using namespace field_config;
using namespace icicle;

#define HIP_FN(name) CONCAT_EXPAND(CONCAT_EXPAND(icicle_hip, ICICLE_FFI_PREFIX), name)

static_assert(sizeof(VecOpsConfig) == sizeof(hip_vec_ops_config_t), "VecOpsConfig layout drifted");
// scalar Montgomery conversion (icicle/include/icicle/backend/vec_ops_backend.h:144-151)
static eIcicleError hip_scalar_convert(const Device& device, const scalar_t* input, uint64_t size, bool is_to_montgomery, const VecOpsConfig& config, scalar_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_vec_ops_config_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(scalar_convert_montgomery)(input, size, is_to_montgomery, &c, output);
}
REGISTER_CONVERT_MONTGOMERY_BACKEND("HIP", hip_scalar_convert);

// element-wise vector ops next to the NTT (icicle/include/icicle/backend/vec_ops_backend.h:25-31,87-132,216-222)
#define HIP_VEC2(NAME)                                                                                                 \
  static eIcicleError hip_##NAME(const Device& device, const scalar_t* a, const scalar_t* b, uint64_t size, const VecOpsConfig& config, scalar_t* output) \
  {                                                                                                                    \
    if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;                                    \
    hip_vec_ops_config_t c;                                                                                            \
    std::memcpy(&c, &config, sizeof(c));                                                                               \
    c.ext = nullptr;                                                                                                   \
    return (eIcicleError)HIP_FN(NAME)(a, b, size, &c, output);                                                         \
  }
HIP_VEC2(vector_add)
HIP_VEC2(vector_sub)
HIP_VEC2(vector_mul)
HIP_VEC2(scalar_mul_vec)
HIP_VEC2(scalar_add_vec)
HIP_VEC2(scalar_sub_vec)
static eIcicleError hip_bit_reverse(const Device& device, const scalar_t* input, uint64_t size, const VecOpsConfig& config, scalar_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_vec_ops_config_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(bit_reverse)(input, size, &c, output);
}
// vector_accumulate(a, b): a[i] += b[i] in place (include/icicle/backend/vec_ops_backend.h) = vector_add with output = a
static eIcicleError hip_vector_accumulate(const Device& device, scalar_t* a, const scalar_t* b, uint64_t size, const VecOpsConfig& config)
{
  VecOpsConfig c = config;
  c.is_result_on_device = c.is_a_on_device;
  return hip_vector_add(device, a, b, size, c, a);
}
REGISTER_VECTOR_ACCUMULATE_BACKEND("HIP", hip_vector_accumulate);
REGISTER_VECTOR_ADD_BACKEND("HIP", hip_vector_add);
REGISTER_VECTOR_SUB_BACKEND("HIP", hip_vector_sub);
REGISTER_VECTOR_MUL_BACKEND("HIP", hip_vector_mul);
REGISTER_SCALAR_MUL_VEC_BACKEND("HIP", hip_scalar_mul_vec);
REGISTER_SCALAR_ADD_VEC_BACKEND("HIP", hip_scalar_add_vec);
REGISTER_SCALAR_SUB_VEC_BACKEND("HIP", hip_scalar_sub_vec);
REGISTER_BIT_REVERSE_BACKEND("HIP", hip_bit_reverse);

// matrix_transpose (icicle/include/icicle/backend/vec_ops_backend.h:33-39,204-212; src/matrix_ops.cpp:73-85): the Rust NTT
// suite transposes on the MAIN device around every columns_batch transform (wrappers/rust/icicle-core/src/ntt/tests.rs:311-335)
static eIcicleError hip_matrix_transpose(const Device& device, const scalar_t* in, uint32_t nof_rows, uint32_t nof_cols, const VecOpsConfig& config, scalar_t* out)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_vec_ops_config_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(matrix_transpose)(in, nof_rows, nof_cols, &c, out);
}
REGISTER_MATRIX_TRANSPOSE_BACKEND("HIP", hip_matrix_transpose);
#ifdef EXT_FIELD
static eIcicleError hip_ext_matrix_transpose(const Device& device, const extension_t* in, uint32_t nof_rows, uint32_t nof_cols, const VecOpsConfig& config, extension_t* out)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_vec_ops_config_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(extension_matrix_transpose)(in, nof_rows, nof_cols, &c, out);
}
REGISTER_MATRIX_TRANSPOSE_EXT_FIELD_BACKEND("HIP", hip_ext_matrix_transpose);
#endif

#ifdef NTT // (Grumpkin's scalar field has none: icicle/cmake/features.cmake:19)
#ifdef HIP_PLUGIN_SCALAR_FIELD_256
typedef hip_ntt_config_u256_t hip_ntt_config_t;
static_assert(sizeof(scalar_t) == 32, "HIP_PLUGIN_SCALAR_FIELD_256 is for the curves' scalar fields");
#elif defined(HIP_PLUGIN_SCALAR_FIELD_64)
typedef hip_ntt_config_u64_t hip_ntt_config_t;
static_assert(sizeof(scalar_t) == 8, "HIP_PLUGIN_SCALAR_FIELD_64 is for goldilocks");
#else
typedef hip_ntt_config_u32_t hip_ntt_config_t;
static_assert(sizeof(scalar_t) == 4, "31-bit field expected");
#endif
static_assert(sizeof(NTTConfig<scalar_t>) == sizeof(hip_ntt_config_t), "NTTConfig layout drifted");
static_assert(sizeof(NTTInitDomainConfig) == sizeof(hip_ntt_init_domain_config_t), "NTTInitDomainConfig layout drifted");

static hip_ntt_config_t translate(const NTTConfig<scalar_t>& c)
{
  hip_ntt_config_t o;
  std::memcpy(&o, &c, sizeof(o));
  o.ext = nullptr;
  return o;
}

static eIcicleError hip_ntt(const Device& device, const scalar_t* input, int size, NTTDir dir, const NTTConfig<scalar_t>& config, scalar_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_ntt_config_t c = translate(config);
  HipExt ext(config.ext); // "hip_num_devices": batch rows over devices
  c.ext = ext.h;
  return (eIcicleError)HIP_FN(ntt)((const uint32_t*)input, size, (int)dir, &c, (uint32_t*)output);
}

#ifndef HIP_PLUGIN_SCALAR_FIELD_256
static eIcicleError hip_ext_ntt(const Device& device, const extension_t* input, int size, NTTDir dir, const NTTConfig<scalar_t>& config, extension_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  const hip_ntt_config_t c = translate(config);
  return (eIcicleError)HIP_FN(extension_ntt)((const uint32_t*)input, size, (int)dir, &c, (uint32_t*)output);
}
#endif

static eIcicleError hip_ntt_init_domain(const Device& device, const scalar_t& primitive_root, const NTTInitDomainConfig& config)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_ntt_init_domain_config_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(ntt_init_domain)((const uint32_t*)&primitive_root, &c);
}

static eIcicleError hip_ntt_release_domain(const Device& device, const scalar_t&)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  return (eIcicleError)HIP_FN(ntt_release_domain)();
}

static eIcicleError hip_get_rou_from_domain(const Device& device, uint64_t logn, scalar_t* rou)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  return (eIcicleError)HIP_FN(get_root_of_unity_from_domain)(logn, (uint32_t*)rou);
}

REGISTER_NTT_INIT_DOMAIN_BACKEND("HIP", hip_ntt_init_domain);
REGISTER_NTT_RELEASE_DOMAIN_BACKEND("HIP", hip_ntt_release_domain);
REGISTER_NTT_GET_ROU_FROM_DOMAIN_BACKEND("HIP", hip_get_rou_from_domain);
REGISTER_NTT_BACKEND("HIP", hip_ntt);
#ifndef HIP_PLUGIN_SCALAR_FIELD_256
REGISTER_NTT_EXT_FIELD_BACKEND("HIP", hip_ext_ntt);

static eIcicleError hip_ext_scalar_convert(const Device& device, const extension_t* input, uint64_t size, bool is_to_montgomery, const VecOpsConfig& config, extension_t* output)
{
  if (icicle_hip_set_device(device.id) != 0) return eIcicleError::INVALID_DEVICE;
  hip_vec_ops_config_t c;
  std::memcpy(&c, &config, sizeof(c));
  c.ext = nullptr;
  return (eIcicleError)HIP_FN(extension_scalar_convert_montgomery)(input, size, is_to_montgomery, &c, output);
}
REGISTER_CONVERT_MONTGOMERY_EXT_FIELD_BACKEND("HIP", hip_ext_scalar_convert);
#endif // !HIP_PLUGIN_SCALAR_FIELD_256
#endif // NTT
