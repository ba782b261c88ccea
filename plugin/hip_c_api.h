// Prototypes of the collision-free libicicle_hip.so entry points the plugin forwards to. This is the
// subset of include/icicle_hip.h that can coexist with the reference headers in one translation unit
// (the full header re-declares the icicle_* runtime names with C types).
#pragma once
#include <stdint.h>

extern "C" {
typedef struct {
  void* stream;
  int precompute_factor, c, bitsize, batch_size;
  bool are_points_shared_in_batch, are_scalars_on_device, are_scalars_montgomery_form, are_points_on_device,
    are_points_montgomery_form, are_results_on_device, is_async;
  void* ext;
} hip_msm_config_t; // == icicle_msm_config_t == icicle::MSMConfig (40 bytes)

typedef struct {
  void* stream;
  uint32_t coset_gen;
  int batch_size;
  bool columns_batch;
  int ordering;
  bool are_inputs_on_device, are_outputs_on_device, is_async;
  void* ext;
} hip_ntt_config_u32_t; // == icicle_ntt_config_u32_t == icicle::NTTConfig<4-byte S> (40 bytes)

typedef struct {
  void* stream;
  uint32_t coset_gen[8];
  int batch_size;
  bool columns_batch;
  int ordering;
  bool are_inputs_on_device, are_outputs_on_device, is_async;
  void* ext;
} hip_ntt_config_u256_t; // == icicle_ntt_config_u256_t == icicle::NTTConfig<32-byte scalar_t> (64 bytes)

typedef struct {
  void* stream;
  uint32_t coset_gen[2];
  int batch_size;
  bool columns_batch;
  int ordering;
  bool are_inputs_on_device, are_outputs_on_device, is_async;
  void* ext;
} hip_ntt_config_u64_t; // == icicle_ntt_config_u64_t == icicle::NTTConfig<goldilocks::scalar_t> (40 bytes)

typedef struct {
  void* stream;
  bool is_async;
  void* ext;
} hip_ntt_init_domain_config_t;

typedef struct {
  void* stream;
  bool is_a_on_device, is_b_on_device, is_result_on_device, is_async;
  int batch_size;
  bool columns_batch;
  void* ext;
} hip_vec_ops_config_t; // == icicle_vec_ops_config_t == icicle::VecOpsConfig (32 bytes)

int icicle_hip_set_device(int device_id);
void* icicle_hip_create_config_extension(void);
void icicle_hip_destroy_config_extension(void* ext);
void icicle_hip_config_extension_set_int(void* ext, const char* key, int value);
void icicle_hip_config_extension_set_bool(void* ext, const char* key, bool value);
#define HIP_DECLARE_CONVERT(P)                                                                                         \
  int icicle_hip_##P##_scalar_convert_montgomery(const void*, uint64_t, bool, const hip_vec_ops_config_t*, void*);
HIP_DECLARE_CONVERT(bn254)
HIP_DECLARE_CONVERT(bls12_381)
HIP_DECLARE_CONVERT(bls12_377)
HIP_DECLARE_CONVERT(grumpkin)
HIP_DECLARE_CONVERT(stark252)
HIP_DECLARE_CONVERT(babybear)
HIP_DECLARE_CONVERT(koalabear)
int icicle_hip_babybear_extension_scalar_convert_montgomery(const void*, uint64_t, bool, const hip_vec_ops_config_t*, void*);
int icicle_hip_koalabear_extension_scalar_convert_montgomery(const void*, uint64_t, bool, const hip_vec_ops_config_t*, void*);
#define HIP_DECLARE_POINT_CONVERT(C)                                                                                   \
  int icicle_hip_##C##_affine_convert_montgomery(const void*, uint64_t, bool, const hip_vec_ops_config_t*, void*);     \
  int icicle_hip_##C##_projective_convert_montgomery(const void*, uint64_t, bool, const hip_vec_ops_config_t*, void*);
HIP_DECLARE_POINT_CONVERT(bn254)
HIP_DECLARE_POINT_CONVERT(bls12_381)
HIP_DECLARE_POINT_CONVERT(bls12_377)
HIP_DECLARE_POINT_CONVERT(grumpkin)
#define HIP_DECLARE_CURVE(C)                                                                                           \
  int icicle_hip_##C##_msm(const void*, const void*, int, const hip_msm_config_t*, void*);                              \
  int icicle_hip_##C##_msm_precompute_bases(const void*, int, const hip_msm_config_t*, void*);
HIP_DECLARE_CURVE(bn254)
HIP_DECLARE_CURVE(bls12_381)
HIP_DECLARE_CURVE(bls12_377)
HIP_DECLARE_CURVE(grumpkin)
HIP_DECLARE_CURVE(bn254_g2)
HIP_DECLARE_CURVE(bls12_381_g2)
HIP_DECLARE_CURVE(bls12_377_g2)
HIP_DECLARE_POINT_CONVERT(bn254_g2)
HIP_DECLARE_POINT_CONVERT(bls12_381_g2)
HIP_DECLARE_POINT_CONVERT(bls12_377_g2)
#define HIP_DECLARE_FIELD(F)                                                                                           \
  int icicle_hip_##F##_ntt(const uint32_t*, int, int, const hip_ntt_config_u32_t*, uint32_t*);                         \
  int icicle_hip_##F##_extension_ntt(const uint32_t*, int, int, const hip_ntt_config_u32_t*, uint32_t*);               \
  int icicle_hip_##F##_ntt_init_domain(const uint32_t*, const hip_ntt_init_domain_config_t*);                          \
  int icicle_hip_##F##_ntt_release_domain(void);                                                                       \
  int icicle_hip_##F##_get_root_of_unity_from_domain(uint64_t, uint32_t*);
HIP_DECLARE_FIELD(babybear)
HIP_DECLARE_FIELD(koalabear)
#define HIP_DECLARE_SCALAR_FIELD(F)                                                                                    \
  int icicle_hip_##F##_ntt(const uint32_t*, int, int, const hip_ntt_config_u256_t*, uint32_t*);                        \
  int icicle_hip_##F##_ntt_init_domain(const uint32_t*, const hip_ntt_init_domain_config_t*);                          \
  int icicle_hip_##F##_ntt_release_domain(void);                                                                       \
  int icicle_hip_##F##_get_root_of_unity_from_domain(uint64_t, uint32_t*);
HIP_DECLARE_SCALAR_FIELD(bn254)
HIP_DECLARE_SCALAR_FIELD(bls12_381)
HIP_DECLARE_SCALAR_FIELD(bls12_377)
HIP_DECLARE_SCALAR_FIELD(stark252)
#define HIP_DECLARE_VEC_ARITH(F)                                                                                        \
  int icicle_hip_##F##_vector_add(const void*, const void*, uint64_t, const hip_vec_ops_config_t*, void*);              \
  int icicle_hip_##F##_vector_sub(const void*, const void*, uint64_t, const hip_vec_ops_config_t*, void*);              \
  int icicle_hip_##F##_vector_mul(const void*, const void*, uint64_t, const hip_vec_ops_config_t*, void*);              \
  int icicle_hip_##F##_scalar_mul_vec(const void*, const void*, uint64_t, const hip_vec_ops_config_t*, void*);          \
  int icicle_hip_##F##_scalar_add_vec(const void*, const void*, uint64_t, const hip_vec_ops_config_t*, void*);          \
  int icicle_hip_##F##_scalar_sub_vec(const void*, const void*, uint64_t, const hip_vec_ops_config_t*, void*);          \
  int icicle_hip_##F##_bit_reverse(const void*, uint64_t, const hip_vec_ops_config_t*, void*);
HIP_DECLARE_VEC_ARITH(babybear)
HIP_DECLARE_VEC_ARITH(koalabear)
HIP_DECLARE_VEC_ARITH(bn254)
HIP_DECLARE_VEC_ARITH(bls12_381)
HIP_DECLARE_VEC_ARITH(bls12_377)
HIP_DECLARE_VEC_ARITH(grumpkin)
HIP_DECLARE_VEC_ARITH(stark252)
HIP_DECLARE_VEC_ARITH(goldilocks)
#define HIP_DECLARE_TRANSPOSE(F, S) int icicle_hip_##F##_##S(const void*, uint32_t, uint32_t, const hip_vec_ops_config_t*, void*);
HIP_DECLARE_TRANSPOSE(babybear, matrix_transpose)
HIP_DECLARE_TRANSPOSE(koalabear, matrix_transpose)
HIP_DECLARE_TRANSPOSE(babybear, extension_matrix_transpose)
HIP_DECLARE_TRANSPOSE(koalabear, extension_matrix_transpose)
HIP_DECLARE_TRANSPOSE(goldilocks, matrix_transpose)
HIP_DECLARE_TRANSPOSE(goldilocks, extension_matrix_transpose)
HIP_DECLARE_TRANSPOSE(bn254, matrix_transpose)
HIP_DECLARE_TRANSPOSE(bls12_381, matrix_transpose)
HIP_DECLARE_TRANSPOSE(bls12_377, matrix_transpose)
HIP_DECLARE_TRANSPOSE(grumpkin, matrix_transpose)
HIP_DECLARE_TRANSPOSE(stark252, matrix_transpose)
HIP_DECLARE_CONVERT(goldilocks)
int icicle_hip_goldilocks_extension_scalar_convert_montgomery(const void*, uint64_t, bool, const hip_vec_ops_config_t*, void*);
int icicle_hip_goldilocks_ntt(const uint32_t*, int, int, const hip_ntt_config_u64_t*, uint32_t*);
int icicle_hip_goldilocks_extension_ntt(const uint32_t*, int, int, const hip_ntt_config_u64_t*, uint32_t*);
int icicle_hip_goldilocks_ntt_init_domain(const uint32_t*, const hip_ntt_init_domain_config_t*);
int icicle_hip_goldilocks_ntt_release_domain(void);
int icicle_hip_goldilocks_get_root_of_unity_from_domain(uint64_t, uint32_t*);
int icicle_hip_bn254_ecntt(const void*, int, int, const hip_ntt_config_u256_t*, void*);
int icicle_hip_bls12_381_ecntt(const void*, int, int, const hip_ntt_config_u256_t*, void*);
int icicle_hip_bls12_377_ecntt(const void*, int, int, const hip_ntt_config_u256_t*, void*);
}

#ifdef __cplusplus
  #include "icicle/config_extension.h"
// The reference's ConfigExtension (a C++ object of libicicle_device.so) -> this backend's own key bag, for the keys
// this backend reads (include/icicle_hip.h "ConfigExtension"); everything else stays behind, tolerated and ignored.
struct HipExt {
  void* h = nullptr;
  explicit HipExt(const icicle::ConfigExtension* e)
  {
    if (!e) return;
    const bool nd = e->has("hip_num_devices"), xb = e->has("hip_msm_exchange_buckets"), rb = e->has("hip_bases_resident"), fr = e->has("hip_force_rccl");
    const bool bg = e->has("hip_bases_generation"), mw = e->has("hip_msm_windows");
    if (!nd && !xb && !rb && !fr && !bg && !mw) return;
    h = icicle_hip_create_config_extension();
    if (bg) icicle_hip_config_extension_set_int(h, "hip_bases_generation", e->get<int>("hip_bases_generation"));
    if (mw) icicle_hip_config_extension_set_int(h, "hip_msm_windows", e->get<int>("hip_msm_windows"));
    if (nd) icicle_hip_config_extension_set_int(h, "hip_num_devices", e->get<int>("hip_num_devices"));
    if (xb) icicle_hip_config_extension_set_bool(h, "hip_msm_exchange_buckets", e->get<bool>("hip_msm_exchange_buckets"));
    if (rb) icicle_hip_config_extension_set_bool(h, "hip_bases_resident", e->get<bool>("hip_bases_resident"));
    if (fr) icicle_hip_config_extension_set_bool(h, "hip_force_rccl", e->get<bool>("hip_force_rccl"));
  }
  ~HipExt()
  {
    if (h) icicle_hip_destroy_config_extension(h);
  }
  HipExt(const HipExt&) = delete;
  HipExt& operator=(const HipExt&) = delete;
};
#endif
