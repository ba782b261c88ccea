/*
 * icicle_hip.h -- C ABI of libicicle_hip.so, the MI355X (gfx950) backend for ICICLE's MSM / NTT hot path.
 *
 * Every entry point below has the NAME, ARGUMENT ORDER, STRUCT LAYOUT and ERROR CODES of the
 * extern "C" symbol the reference's Rust / Go wrappers bind (wrappers/rust/icicle-core/src/msm/mod.rs:249-266,
 * ntt/mod.rs:285-293,346-355, icicle-runtime/src/runtime.rs:10-54; wrappers/golang/curves/bn254/msm/include/msm.h:15-16),
 * so those bindings link against this library unchanged. Each declaration cites the reference
 * definition it replaces (paths relative to /root/reference/icicle).
 *
 * Plain C: pointers and sizes only. C++ references in the reference's extern "C" signatures
 * (`const Device&`, `int&`) are passed as pointers here -- the same machine ABI.
 */
#ifndef ICICLE_HIP_H
#define ICICLE_HIP_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
  #include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- errors: include/icicle/errors.h:13-29. Only 0..11 are ever returned (Rust's enum stops at 12). ---- */
typedef enum {
  ICICLE_SUCCESS = 0,
  ICICLE_INVALID_DEVICE = 1,
  ICICLE_OUT_OF_MEMORY = 2,
  ICICLE_INVALID_POINTER = 3,
  ICICLE_ALLOCATION_FAILED = 4,
  ICICLE_DEALLOCATION_FAILED = 5,
  ICICLE_COPY_FAILED = 6,
  ICICLE_SYNCHRONIZATION_FAILED = 7,
  ICICLE_STREAM_CREATION_FAILED = 8,
  ICICLE_STREAM_DESTRUCTION_FAILED = 9,
  ICICLE_API_NOT_IMPLEMENTED = 10,
  ICICLE_INVALID_ARGUMENT = 11
} icicle_error_t;

/* ---- device: include/icicle/device.h:14-48 (sizeof 68, id at offset 64) ---- */
typedef struct {
  char type[64]; /* "HIP" */
  int id;        /* GPU ordinal */
} icicle_device_t;

/* include/icicle/device.h:53-58 */
typedef struct {
  bool using_host_memory;
  int num_memory_regions;
  bool supports_pinned_memory;
} icicle_device_properties_t;

typedef void* icicleStreamHandle; /* a hipStream_t; NULL = default stream (include/icicle/runtime.h) */

/* ======================================================================================
 * Runtime: include/icicle/runtime.h:17-281, src/runtime.cpp.
 * The only registered device type is "HIP". Pointers returned by icicle_malloc* are recorded in an
 * allocation tracker (include/icicle/memory_tracker.h:16-44) so icicle_copy can infer direction.
 * ====================================================================================== */
icicle_error_t icicle_load_backend(const char* path, bool is_recursive); /* runtime.h:17  (no-op: backend is linked in) */
icicle_error_t icicle_load_backend_from_env_or_default(void);            /* runtime.h:29  (no-op) */
icicle_error_t icicle_set_device(const icicle_device_t* device);         /* runtime.h:37 */
icicle_error_t icicle_set_default_device(const icicle_device_t* device); /* runtime.h:45 */
icicle_error_t icicle_get_active_device(icicle_device_t* device);        /* runtime.h:53 */
icicle_error_t icicle_is_host_memory(const void* ptr);                   /* runtime.h:61 */
icicle_error_t icicle_is_active_device_memory(const void* ptr);          /* runtime.h:69 */
icicle_error_t icicle_get_device_count(int* device_count);               /* runtime.h:77 */
icicle_error_t icicle_malloc(void** ptr, size_t size);                   /* runtime.h:86 */
icicle_error_t icicle_malloc_async(void** ptr, size_t size, icicleStreamHandle stream); /* runtime.h:97 */
icicle_error_t icicle_free(void* ptr);                                   /* runtime.h:105 */
icicle_error_t icicle_free_async(void* ptr, icicleStreamHandle stream);  /* runtime.h:114 */
icicle_error_t icicle_get_available_memory(size_t* total, size_t* free); /* runtime.h:123 */
icicle_error_t icicle_memset(void* ptr, int value, size_t size);         /* runtime.h:135 */
icicle_error_t icicle_memset_async(void* ptr, int value, size_t size, icicleStreamHandle stream); /* runtime.h:149 */
icicle_error_t icicle_copy(void* dst, const void* src, size_t size);     /* runtime.h:161 */
icicle_error_t icicle_copy_async(void* dst, const void* src, size_t size, icicleStreamHandle stream); /* :172 */
icicle_error_t icicle_copy_to_host(void* dst, const void* src, size_t size);                          /* :184 */
icicle_error_t icicle_copy_to_host_async(void* dst, const void* src, size_t size, icicleStreamHandle stream); /* :195 */
icicle_error_t icicle_copy_to_device(void* dst, const void* src, size_t size);                        /* :205 */
icicle_error_t icicle_copy_to_device_async(void* dst, const void* src, size_t size, icicleStreamHandle stream); /* :216 */
icicle_error_t icicle_create_stream(icicleStreamHandle* stream);         /* runtime.h:227 */
icicle_error_t icicle_destroy_stream(icicleStreamHandle stream);         /* runtime.h:236 */
icicle_error_t icicle_stream_synchronize(icicleStreamHandle stream);     /* runtime.h:246 */
icicle_error_t icicle_device_synchronize(void);                          /* runtime.h:253 */
icicle_error_t icicle_get_device_properties(icicle_device_properties_t* properties); /* runtime.h:261 */
icicle_error_t icicle_is_device_available(const icicle_device_t* dev);   /* runtime.h:271 */
icicle_error_t icicle_get_registered_devices(char* output, size_t output_size); /* runtime.h:281 */

/* ---- ConfigExtension: src/config_extension.cpp:7-37 (string-keyed int/bool bag, opaque handle) ----
 * The reference hands backend-specific knobs to a backend through MSMConfig.ext / NTTConfig.ext
 * (include/icicle/backend/msm_config.h:4-16, ntt_config.h:9-18). Keys read by this backend:
 *   "hip_num_devices"          (int,  msm + ntt)  G >= 1: run the call on G shards over min(G, visible GPUs) devices
 *                                                 starting at the active one (MSM: contiguous (scalar, base) shards,
 *                                                 partial results all-gathered over RCCL and summed; batched NTT:
 *                                                 rows per device, no collective). icicle_amd/csrc/msm_multi.hpp.
 *   "hip_msm_exchange_buckets" (bool, msm)        with hip_num_devices: exchange bucket slices (all-to-all) instead of
 *                                                 final partial sums, so that the bucket reduction is sharded too.
 * Foreign keys (the CUDA backend's "large_bucket_factor", "fast_twiddles", the CPU backend's "n_threads", ...) are
 * tolerated and ignored. */
typedef struct icicle_config_extension icicle_config_extension_t;
icicle_config_extension_t* create_config_extension(void);
void destroy_config_extension(icicle_config_extension_t* ext);
void config_extension_set_int(icicle_config_extension_t* ext, const char* key, int value);
void config_extension_set_bool(icicle_config_extension_t* ext, const char* key, bool value);
int config_extension_get_int(const icicle_config_extension_t* ext, const char* key);
bool config_extension_get_bool(const icicle_config_extension_t* ext, const char* key);
icicle_config_extension_t* clone_config_extension(const icicle_config_extension_t* ext);

/* ======================================================================================
 * MSM: include/icicle/msm.h:21-53 (MSMConfig, 40 bytes), src/msm.cpp:12-16, 45-49.
 * scalars: batch*N canonical (or reference-Montgomery) scalar_t = 8 x u32 little-endian.
 * bases:   affine_t {x,y}, 2*L x u32 (L = 8 bn254, 12 bls12_381); identity = (0,0).
 * results: batch projective_t {x,y,z}, 3*L x u32, canonical; identity = (0:1:0).
 * ====================================================================================== */
typedef struct {
  icicleStreamHandle stream;       /* offset 0  */
  int precompute_factor;           /* 8  */
  int c;                           /* 12 */
  int bitsize;                     /* 16 */
  int batch_size;                  /* 20 */
  bool are_points_shared_in_batch; /* 24 */
  bool are_scalars_on_device;      /* 25 */
  bool are_scalars_montgomery_form;/* 26 */
  bool are_points_on_device;       /* 27 */
  bool are_points_montgomery_form; /* 28 */
  bool are_results_on_device;      /* 29 */
  bool is_async;                   /* 30 */
  icicle_config_extension_t* ext;  /* 32 */
} icicle_msm_config_t;

/* <curve>_msm_precompute_bases: `nof_bases` is the number of bases of ONE MSM, as both wrappers of the reference pass it
 * (wrappers/rust/icicle-core/src/msm/mod.rs:296, wrappers/golang/curves/bn254/msm/msm.go:39-43): with per-MSM bases
 * (batch_size > 1, are_points_shared_in_batch = false) input_bases holds nof_bases * batch_size points and output_bases
 * takes precompute_factor times as many; layout output[precompute_factor * i + j] = 2^(j * shift) * input[i]
 * (backend/cpu/src/curve/cpu_msm.hpp:470-485). The output is device memory whenever the pointer says so, even with
 * are_results_on_device left false (the Rust wrapper never sets it for this call).
 * Window size with a table: pass the same config.c > 0 to both calls, or leave both at 0 -- then msm() takes the c the table
 * was built with from a per-process record of the tables msm_precompute_bases wrote (any aligned slice of a device-resident
 * table works, a host-resident one is recognised at its start; whatever msm_size / batch_size), and only for a table it has never seen (copied, loaded from disk) derives c from
 * msm_size, scalar bits and precompute_factor exactly as msm_precompute_bases does from nof_bases -- the reference's rule
 * (cpu_msm.hpp:466 vs :207), correct when the two sizes are equal. */
icicle_error_t bn254_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results); /* src/msm.cpp:12 */
icicle_error_t bn254_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases); /* src/msm.cpp:45 */
icicle_error_t bls12_381_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results);
icicle_error_t bls12_381_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases);
/* G2 (the reference's G2_ENABLED build): bases are g2_affine_t = {x.c0, x.c1, y.c0, y.c1} over the base field
 * (curves/params/bn254.h:15-17, fields/complex_extension.h), results g2_projective_t = {x, y, z}. Same MSMConfig. */
icicle_error_t bn254_g2_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results); /* src/msm.cpp:28 */
icicle_error_t bn254_g2_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases); /* src/msm.cpp:61 */
icicle_error_t bls12_381_g2_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results);
icicle_error_t bls12_381_g2_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases);

/* ======================================================================================
 * NTT: include/icicle/ntt.h:23-26 (NTTDir), 37-44 (Ordering), 53-64 (NTTConfig<S>), 92-96
 * (NTTInitDomainConfig); src/ntt.cpp:11-84.  NTTConfig<S> for a 4-byte S is 40 bytes.
 * ====================================================================================== */
typedef enum { ICICLE_NTT_FORWARD = 0, ICICLE_NTT_INVERSE = 1 } icicle_ntt_dir_t;
typedef enum { ICICLE_kNN = 0, ICICLE_kNR = 1, ICICLE_kRN = 2, ICICLE_kRR = 3, ICICLE_kNM = 4, ICICLE_kMN = 5 } icicle_ordering_t;

typedef struct {
  icicleStreamHandle stream;      /* 0  */
  uint32_t coset_gen;             /* 8   canonical field element, 1 = no coset */
  int batch_size;                 /* 12 */
  bool columns_batch;             /* 16 */
  int ordering;                   /* 20  icicle_ordering_t */
  bool are_inputs_on_device;      /* 24 */
  bool are_outputs_on_device;     /* 25 */
  bool is_async;                  /* 26 */
  icicle_config_extension_t* ext; /* 32 */
} icicle_ntt_config_u32_t;

typedef struct {
  icicleStreamHandle stream;      /* 0 */
  bool is_async;                  /* 8 */
  icicle_config_extension_t* ext; /* 16 */
} icicle_ntt_init_domain_config_t;

#define ICICLE_HIP_DECLARE_NTT_U32(F)                                                                                  \
  icicle_error_t F##_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* config, uint32_t* output); /* src/ntt.cpp:11 */ \
  icicle_error_t F##_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config);  /* src/ntt.cpp:26 */ \
  icicle_error_t F##_ntt_release_domain(void);                                                                          /* src/ntt.cpp:41 */ \
  icicle_error_t F##_get_root_of_unity(uint64_t max_size, uint32_t* rou);                                               /* src/ntt.cpp:55 */ \
  icicle_error_t F##_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou);                                       /* src/ntt.cpp:75 */ \
  icicle_error_t F##_extension_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* config, uint32_t* output); /* src/ntt.cpp:90 */

ICICLE_HIP_DECLARE_NTT_U32(babybear)
ICICLE_HIP_DECLARE_NTT_U32(koalabear)

/* NTT over the curves' 256-bit scalar fields: scalar_t = 8 little-endian u32 words, so NTTConfig<scalar_t>
 * is 64 bytes (include/icicle/ntt.h:53-64 with S = bn254::scalar_t; storage<8> is 8-byte aligned on the host,
 * include/icicle/math/storage.h:36-49). Symbols: src/ntt.cpp:11-84 built with FIELD = the curve's scalar field. */
typedef struct {
  icicleStreamHandle stream;
  uint32_t coset_gen[8]; /* canonical, {1,0,..} = no coset */
  int32_t batch_size;
  bool columns_batch;
  int32_t ordering; /* icicle_ntt_ordering_t */
  bool are_inputs_on_device;
  bool are_outputs_on_device;
  bool is_async;
  icicle_config_extension_t* ext;
} icicle_ntt_config_u256_t;

#define ICICLE_HIP_DECLARE_NTT_U256(F)                                                                                 \
  icicle_error_t F##_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u256_t* config, uint32_t* output); /* src/ntt.cpp:11 */ \
  icicle_error_t F##_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config);  /* src/ntt.cpp:26 */ \
  icicle_error_t F##_ntt_release_domain(void);                                                                          /* src/ntt.cpp:41 */ \
  icicle_error_t F##_get_root_of_unity(uint64_t max_size, uint32_t* rou);                                               /* src/ntt.cpp:55 */ \
  icicle_error_t F##_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou);                                       /* src/ntt.cpp:75 */

ICICLE_HIP_DECLARE_NTT_U256(bn254)
ICICLE_HIP_DECLARE_NTT_U256(bls12_381)
/* ECNTT: NTT over G1 points with scalar-field twiddles (src/ecntt.cpp:7-11). input/output: projective_t[size*batch]
 * (3 base-field elements each); config and twiddle domain are those of <curve>_ntt. */
icicle_error_t bn254_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output);
icicle_error_t bls12_381_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output);

/* ======================================================================================
 * Montgomery-form conversion (vec-ops): include/icicle/vec_ops.h:19-37 (VecOpsConfig, 32 bytes),
 * src/vec_ops.cpp:404-408,421-425 (<prefix>_scalar_convert_montgomery, _extension_scalar_convert_montgomery),
 * src/curves/montgomery_conversion.cpp:12-16,46-50 (<curve>_affine/_projective_convert_montgomery).
 * x -> x*R (is_to_montgomery) or x*R^-1 mod p, R = 2^(32*limbs); element-wise, any layout.
 * ====================================================================================== */
typedef struct {
  icicleStreamHandle stream;      /* 0  */
  bool is_a_on_device;            /* 8  */
  bool is_b_on_device;            /* 9  */
  bool is_result_on_device;       /* 10 */
  bool is_async;                  /* 11 */
  int batch_size;                 /* 12 */
  bool columns_batch;             /* 16 */
  icicle_config_extension_t* ext; /* 24 */
} icicle_vec_ops_config_t;

#define ICICLE_HIP_DECLARE_CONVERT(P)                                                                                  \
  icicle_error_t P##_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
ICICLE_HIP_DECLARE_CONVERT(bn254)
ICICLE_HIP_DECLARE_CONVERT(bls12_381)
ICICLE_HIP_DECLARE_CONVERT(babybear)
ICICLE_HIP_DECLARE_CONVERT(koalabear)
icicle_error_t babybear_extension_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t koalabear_extension_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t bn254_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t bn254_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t bls12_381_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t bls12_381_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t bn254_g2_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output); /* src/curves/montgomery_conversion.cpp:29 */
icicle_error_t bn254_g2_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output); /* :63 */
icicle_error_t bls12_381_g2_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t bls12_381_g2_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);

/* Element-wise vector operations next to the NTT path, for every NTT field (src/vec_ops.cpp:71-84 vector_add,
 * :136-149 vector_sub, :169-182 vector_mul, :362-366 scalar_mul_vec -- one scalar per batch entry --, :440-444
 * bit_reverse). `size` is per batch entry; config.batch_size / columns_batch as in the reference. */
#define ICICLE_HIP_DECLARE_VEC_ARITH(F)                                                                                \
  icicle_error_t F##_vector_add(const void* vec_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t F##_vector_sub(const void* vec_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t F##_vector_mul(const void* vec_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t F##_scalar_mul_vec(const void* scalar_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t F##_scalar_add_vec(const void* scalar_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t F##_scalar_sub_vec(const void* scalar_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t icicle_hip_##F##_scalar_add_vec(const void* scalar_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t icicle_hip_##F##_scalar_sub_vec(const void* scalar_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t F##_bit_reverse(const void* input, uint64_t size, const icicle_vec_ops_config_t* config, void* output); \
  icicle_error_t icicle_hip_##F##_vector_add(const void* vec_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t icicle_hip_##F##_vector_sub(const void* vec_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t icicle_hip_##F##_vector_mul(const void* vec_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t icicle_hip_##F##_scalar_mul_vec(const void* scalar_a, const void* vec_b, uint64_t size, const icicle_vec_ops_config_t* config, void* result); \
  icicle_error_t icicle_hip_##F##_bit_reverse(const void* input, uint64_t size, const icicle_vec_ops_config_t* config, void* output);
ICICLE_HIP_DECLARE_VEC_ARITH(babybear)
ICICLE_HIP_DECLARE_VEC_ARITH(koalabear)
ICICLE_HIP_DECLARE_VEC_ARITH(bn254)
ICICLE_HIP_DECLARE_VEC_ARITH(bls12_381)

/* Matrix transpose of batch_size row-major nof_rows x nof_cols matrices (icicle/src/matrix_ops.cpp:75-102,
 * include/icicle/vec_ops.h:319; CPU: backend/cpu/src/field/cpu_matrix_ops.cpp:333-362): in place allowed, columns_batch
 * rejected. The Rust NTT suite calls it on the main device around every columns_batch transform
 * (wrappers/rust/icicle-core/src/ntt/tests.rs:311-335). */
#define ICICLE_HIP_DECLARE_TRANSPOSE(F, S)                                                                             \
  icicle_error_t F##_##S(const void* mat_in, uint32_t nof_rows, uint32_t nof_cols, const icicle_vec_ops_config_t* config, void* mat_out); \
  icicle_error_t icicle_hip_##F##_##S(const void* mat_in, uint32_t nof_rows, uint32_t nof_cols, const icicle_vec_ops_config_t* config, void* mat_out);
ICICLE_HIP_DECLARE_TRANSPOSE(babybear, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(koalabear, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(babybear, extension_matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(koalabear, extension_matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(goldilocks, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(goldilocks, extension_matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(bn254, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(bls12_381, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(bls12_377, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(grumpkin, matrix_transpose)
ICICLE_HIP_DECLARE_TRANSPOSE(stark252, matrix_transpose)

/* ---- backend-specific helpers (not part of the reference ABI) ---- */
const char* icicle_hip_version(void);
/* The window plan msm() would use for this size / config: c = window bits, nwin = number of c-bit windows of a
 * scalar_bits-bit scalar (signed digits, so ceil((scalar_bits + 1) / c)). For operation counts in benchmarks. */
icicle_error_t icicle_hip_msm_plan(int msm_size, int scalar_bits, const icicle_msm_config_t* config, int* c, int* nwin);
/* The whole plan, for tests and benchmarks: out[0..7] = c (bits of the widest windows), nwin, n_lo (number of windows that are one bit
 * narrower -- the mixed-width plans picked from 2^23 terms up), negate (1 = scalars with the top bit set are replaced by r - s and
 * their point negated, icicle/backend/cpu/src/curve/cpu_msm.hpp:276-277), buckets per window slot, buckets in use, segment size,
 * windows per precomputed-table entry. force_windows = MSMConfig.ext "hip_msm_windows" (0: the cost model decides, as msm() does). */
icicle_error_t icicle_hip_msm_plan_info(int msm_size, int scalar_bits, const icicle_msm_config_t* config, int force_windows, int* out);
/* Device-side synthetic input generator for benchmarks: fills `out` (device or host per flag) with
 * `n` DISTINCT affine points (k0 + i) * G in the reference's canonical affine layout. */
icicle_error_t bn254_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream);
icicle_error_t bls12_381_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream);
/* Device-side sum of n projective points (reference layout, device pointers): the combine step of a
 * base-sharded multi-GPU MSM after the RCCL all-gather of per-GPU partial results. */
icicle_error_t bn254_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream);
icicle_error_t bls12_381_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream);
icicle_error_t bn254_g2_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream);
icicle_error_t bls12_381_g2_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream);
icicle_error_t bn254_g2_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream);
icicle_error_t bls12_381_g2_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream);
/* data[r][c] *= w_N^(+-(row0+r)*c) on device (N = 2^logn_total, w_N from the initialised domain): the inter-step
 * twiddle of a 4-step NTT whose two steps run on different GPUs (icicle_amd/dist.py, all-to-all over RCCL). */
icicle_error_t babybear_hip_twiddle_rows(uint32_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t logn_total, bool inverse, icicleStreamHandle stream);
icicle_error_t koalabear_hip_twiddle_rows(uint32_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t logn_total, bool inverse, icicleStreamHandle stream);
/* Average device time (ms) of the dominant kernel's launches since the last reset, measured with
 * hipEvents on the launch stream (bench.py's live roofline figure). which: 0 = MSM bucket
 * accumulation, 1 = NTT pass kernels, 2 = MSM digit extraction + bucket sort (everything in front of the
 * accumulation), 3 = MSM bucket reduction + window combine (everything behind it). */
icicle_error_t icicle_hip_kernel_timing(int which, bool reset, double* total_ms, int* launches);
icicle_error_t icicle_hip_enable_kernel_timing(bool enable);
/* Roofs of the dominant MSM kernel, measured on the spot (bench.py reports them next to the kernel's own rate):
 * XYZZ mixed additions per second with all operands in registers (curve 0 = bn254, 1 = bls12_381) -- the integer-ALU
 * roof of bucket accumulation -- and random 64-byte gathers per second over a region of `region_bytes` (the access
 * pattern of its base fetch; also the known-byte-count kernel the FETCH_SIZE counter is calibrated on). */
icicle_error_t icicle_hip_ubench_mixed_add(int curve, double* adds_per_second);
icicle_error_t icicle_hip_ubench_gather(uint64_t region_bytes, uint64_t gathers, double* gathers_per_second);
/* VALU-issue roof of the NTT pass kernels, measured on the spot (field 0 = babybear, 1 = koalabear): "pass units" per second, one
 * unit = 16 elements through one 8-stage pass with every operand in registers (two radix-16 register rounds of k_ntt_fast's own
 * butterfly code + the 16 products with the factor behind the pass; no memory, no LDS). A three-pass 2^24-point transform is
 * 3 * 2^20 units per row (reference semantics of the butterflies: icicle/backend/cpu/include/ntt_cpu.h:129-225). */
icicle_error_t icicle_hip_ubench_ntt_pass(int field, double* pass_units_per_second);
/* Device self-test of the in-place asm field products with aliased and constant operands (curve 0 = bn254, 1 = bls12_381):
 * *mismatches must be 0 (tests/test_gpu_msm.py). */
icicle_error_t icicle_hip_selftest_inplace_products(int curve, int* mismatches);
/* Quad-cooperative complete addition / doubling (the ECNTT's butterflies, the MSM's window combine on the GPU) against the one-lane
 * formulas of the reference (projective.h:73-143) on every pair (a G, +- b G), a, b = 0..6 -- P = Q, P = -Q and the identity included --
 * as group elements. curve 0 = bn254, 1 = bls12_381, 2 = bls12_377; *mismatches must come back 0. */
icicle_error_t icicle_hip_selftest_quad_group_ops(int curve, int* mismatches);
/* msm()/ntt() keep their temporaries cached between calls (about 8 GiB after a 2^26-term MSM). release_workspace gives
 * the idle part back to the device (it is also given back automatically when an allocation would otherwise fail, and by
 * <field>_ntt_release_domain); workspace_bytes reports what is cached for the active device. */
icicle_error_t icicle_hip_release_workspace(void);
icicle_error_t icicle_hip_workspace_bytes(size_t* bytes);
/* MSMConfig.ext / NTTConfig.ext keys this backend reads (include/icicle/backend/msm_config.h:4-16 is the reference's
 * vehicle for backend knobs; the plugin copies them out of the reference's ConfigExtension key by key):
 *   "hip_num_devices"          int   msm + ntt: cut the work into this many shards over min(G, visible GPUs) devices
 *   "hip_msm_exchange_buckets" bool  msm: exchange bucket slices (grouped send / recv) instead of partial results
 *   "hip_bases_resident"       bool  msm: keep the per-device copies of the base shards between calls (the caller
 *                                    promises the bases at that pointer do not change); released by the function below --
 *                                    MANDATORY before the address is reused unless the memory came from icicle_malloc
 *                                    (icicle_free releases the copies) --, or versioned with
 *   "hip_bases_generation"     int   msm: caller's content id, part of the cache key (a new value stages fresh copies)
 *   "hip_msm_windows"          int   msm (single MSM, precompute_factor 1, c = 0, full-width scalars): W windows whose widths add up
 *                                    to the scalar bits, the top ones one bit wider, with the reference's negate-if-top-bit-set trick
 *                                    (cpu_msm.hpp:276-277). The cost model picks such a plan by itself from 2^23 terms up (BN254 2^26:
 *                                    10 x 21 + 2 x 22 bits); the key forces one at any size (parity tests, A/B runs)
 *   "hip_force_rccl"           bool  msm, ntt: take the RCCL exchanges even with one device slot -- all-gather, and the grouped
 *                                    ncclSend / ncclRecv of the bucket exchange and of the split transform as sends to
 *                                    the own rank of a size-1 communicator (test hook: the real librccl on one GPU)
 * icicle_hip_msm_release_resident_bases(bases) frees the resident copies made for `bases` (NULL: for every pointer);
 * icicle_free / icicle_free_async of a device allocation release the copies made for it as well. */
icicle_error_t icicle_hip_msm_release_resident_bases(const void* bases);
/* Counters of what the multi-device / pipelined paths moved since the last reset: { base bytes staged to a device, scalar
 * bytes staged, bytes sent by the bucket exchange / the split transform's all-to-all, resident-base hits (shards NOT staged
 * again), calls that ran one host thread per device slot, point-to-point messages sent, cross-device copies that took the
 * no-peer-access route, MSMs re-run on the uniform window plan after the mixed-width plan did not get its memory }.
 * icicle_hip_multi_stats2 writes the first min(n, 8) of them; icicle_hip_multi_stats writes exactly the first FIVE (out[5], the
 * contract since round 3). */
icicle_error_t icicle_hip_multi_stats(uint64_t* out5, bool reset);
icicle_error_t icicle_hip_multi_stats2(uint64_t* out, int n, bool reset);
/* What the collectives library (RCCL, or the one set with icicle_hip_set_collectives_library) reported for the most recently created
 * communicator set of the in-process multi-GPU path: out[0] ncclGetVersion, out[1] devices asked for, out[2] ncclCommCount,
 * out[3] 1 when every slot p reported ncclCommUserRank == p and ncclCommCount == devices (a set that does not is refused with
 * INVALID_DEVICE), out[4] communicator sets created so far. -1 where the library does not export the call. Writes min(n, 5) values. */
icicle_error_t icicle_hip_collectives_info(int* out, int n);
/* The multi-device paths take their collectives from a library with the NCCL C ABI (ncclCommInitAll, ncclCommDestroy,
 * ncclAllGather, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), bound with dlopen: librccl.so by default,
 * or the library at `path` (NULL / "" = default again; ICICLE_HIP_RCCL_LIB in the environment does the same) -- another RCCL
 * build, or the in-process stand-in the rehearsal tests build from tests/loopback/ (real RCCL refuses one GPU twice in a
 * communicator). Communicators are cached per library and device list. */
icicle_error_t icicle_hip_set_collectives_library(const char* path);
/* Rehearsal hooks for the multi-device code on a box with fewer GPUs than device slots (tests only): K virtual device slots
 * mapped round-robin onto the physical GPUs, and a one-shot failure of device slot `slot` at stage 1 (worker set-up), 2 (right
 * before the bucket-exchange gate / the split transform's second exchange) or 3 (right before the result-gather gate); stage 0
 * disarms. (slot 0, stage 9): the next single-device msm() "fails to allocate" its first plan once -- the rehearsal of the fallback
 * from the auto-selected mixed-width window plan to the uniform one. */
icicle_error_t icicle_hip_test_set_virtual_devices(int slots);
/* Rehearsal of the case "hipDeviceEnablePeerAccess is refused" (also ICICLE_HIP_NO_PEER_ACCESS=1): the in-process multi-GPU paths then move
 * device-resident operands and results with hipMemcpyPeerAsync (staged through host memory by the HIP runtime when the two devices
 * are not peers) instead of direct hipMemcpyDefault copies. Counted in icicle_hip_multi_stats2's 7th value. */
icicle_error_t icicle_hip_test_set_no_peer_access(bool off);
icicle_error_t icicle_hip_test_inject_failure(int slot, int stage);

/* ---- collision-free aliases used by the reference-runtime plugin (plugin/):
 * same functions as the un-prefixed names above, for processes that also load the reference's own
 * libicicle_device / libicicle_curve_<c> / libicicle_field_<f>, which define those names. ---- */
icicle_error_t icicle_hip_set_device(int device_id);
icicle_config_extension_t* icicle_hip_create_config_extension(void);
void icicle_hip_destroy_config_extension(icicle_config_extension_t* ext);
void icicle_hip_config_extension_set_int(icicle_config_extension_t* ext, const char* key, int value);
void icicle_hip_config_extension_set_bool(icicle_config_extension_t* ext, const char* key, bool value);
icicle_error_t icicle_hip_bn254_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results);
icicle_error_t icicle_hip_bn254_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases);
icicle_error_t icicle_hip_bls12_381_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results);
icicle_error_t icicle_hip_bls12_381_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases);
icicle_error_t icicle_hip_bn254_g2_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results);
icicle_error_t icicle_hip_bn254_g2_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases);
icicle_error_t icicle_hip_bls12_381_g2_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results);
icicle_error_t icicle_hip_bls12_381_g2_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases);
#define ICICLE_HIP_DECLARE_NTT_ALIASES(F)                                                                              \
  icicle_error_t icicle_hip_##F##_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* config, uint32_t* output); \
  icicle_error_t icicle_hip_##F##_extension_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* config, uint32_t* output); \
  icicle_error_t icicle_hip_##F##_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config); \
  icicle_error_t icicle_hip_##F##_ntt_release_domain(void);                                                            \
  icicle_error_t icicle_hip_##F##_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou);
ICICLE_HIP_DECLARE_NTT_ALIASES(babybear)
ICICLE_HIP_DECLARE_NTT_ALIASES(koalabear)
#define ICICLE_HIP_DECLARE_NTT_U256_ALIASES(F)                                                                         \
  icicle_error_t icicle_hip_##F##_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u256_t* config, uint32_t* output); \
  icicle_error_t icicle_hip_##F##_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config); \
  icicle_error_t icicle_hip_##F##_ntt_release_domain(void);                                                            \
  icicle_error_t icicle_hip_##F##_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou);
ICICLE_HIP_DECLARE_NTT_U256_ALIASES(bn254)
ICICLE_HIP_DECLARE_NTT_U256_ALIASES(bls12_381)
icicle_error_t icicle_hip_bn254_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output);
icicle_error_t icicle_hip_bls12_381_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output);
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_bn254)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_bls12_381)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_babybear)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_koalabear)
icicle_error_t icicle_hip_babybear_extension_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_koalabear_extension_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bn254_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bn254_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bls12_381_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bls12_381_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bn254_g2_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bn254_g2_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bls12_381_g2_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_bls12_381_g2_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);

/* ======================================================================================
 * Further curves / fields of the reference that have an MSM or an NTT (icicle/cmake/features.cmake:6,17,19), with the
 * signatures and layouts of the bn254 / bls12_381 symbols above:
 *   bls12_377  MSM on G1 (L = 12) and on G2 over Fq2 = Fq[u]/(u^2 + 5), NTT over its 253-bit scalar field, ECNTT,
 *              Montgomery conversion, element-wise vector ops     (curves/params/bls12_377.h)
 *   grumpkin   MSM on G1 (L = 8; base field = BN254's scalar field, scalars = BN254's base field), Montgomery
 *              conversion, vector ops; the reference gives it no NTT  (curves/params/grumpkin.h)
 *   stark252   NTT + Montgomery conversion + vector ops over the 252-bit Stark field 2^251 + 17*2^192 + 1
 *              (fields/stark_fields/stark252.h; no curve)
 *   goldilocks NTT, extension-field NTT (quadratic extension u^2 = 7: the two components are transformed with the base
 *              field's twiddles), Montgomery conversion (x * 2^64), vector ops over 2^64 - 2^32 + 1
 *              (fields/stark_fields/goldilocks.h; scalar_t = 2 words, so NTTConfig<scalar_t> is the 40-byte struct below)
 * bw6_761 (761-bit base field) and m31 (no NTT in the reference) are not built.
 * ====================================================================================== */
typedef struct {
  icicleStreamHandle stream;      /* 0  */
  uint32_t coset_gen[2];          /* 8   canonical, {1,0} = no coset */
  int32_t batch_size;             /* 16 */
  bool columns_batch;             /* 20 */
  int32_t ordering;               /* 24  icicle_ntt_ordering_t */
  bool are_inputs_on_device;      /* 28 */
  bool are_outputs_on_device;     /* 29 */
  bool is_async;                  /* 30 */
  icicle_config_extension_t* ext; /* 32 */
} icicle_ntt_config_u64_t;
#define ICICLE_HIP_DECLARE_MSM(S)                                                                                      \
  icicle_error_t S##_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results); \
  icicle_error_t S##_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases); \
  icicle_error_t icicle_hip_##S##_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results); \
  icicle_error_t icicle_hip_##S##_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases); \
  icicle_error_t S##_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream); \
  icicle_error_t S##_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream);              \
  icicle_error_t S##_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output); \
  icicle_error_t S##_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output); \
  icicle_error_t icicle_hip_##S##_affine_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output); \
  icicle_error_t icicle_hip_##S##_projective_convert_montgomery(const void* input, uint64_t n, bool is_into, const icicle_vec_ops_config_t* config, void* output);
ICICLE_HIP_DECLARE_MSM(bls12_377)
ICICLE_HIP_DECLARE_MSM(bls12_377_g2)
ICICLE_HIP_DECLARE_MSM(grumpkin)
ICICLE_HIP_DECLARE_NTT_U256(bls12_377)
ICICLE_HIP_DECLARE_NTT_U256(stark252)
ICICLE_HIP_DECLARE_NTT_U256_ALIASES(bls12_377)
ICICLE_HIP_DECLARE_NTT_U256_ALIASES(stark252)
icicle_error_t bls12_377_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output);
icicle_error_t icicle_hip_bls12_377_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output);
ICICLE_HIP_DECLARE_CONVERT(bls12_377)
ICICLE_HIP_DECLARE_CONVERT(grumpkin)
ICICLE_HIP_DECLARE_CONVERT(stark252)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_bls12_377)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_grumpkin)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_stark252)
ICICLE_HIP_DECLARE_VEC_ARITH(bls12_377)
ICICLE_HIP_DECLARE_VEC_ARITH(grumpkin)
ICICLE_HIP_DECLARE_VEC_ARITH(stark252)
/* goldilocks: elements are 2 u32 words (extension elements 4) */
icicle_error_t goldilocks_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u64_t* config, uint32_t* output);           /* src/ntt.cpp:11 */
icicle_error_t goldilocks_extension_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u64_t* config, uint32_t* output); /* src/ntt.cpp:90 */
icicle_error_t goldilocks_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config);
icicle_error_t goldilocks_ntt_release_domain(void);
icicle_error_t goldilocks_get_root_of_unity(uint64_t max_size, uint32_t* rou);
icicle_error_t goldilocks_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou);
icicle_error_t icicle_hip_goldilocks_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u64_t* config, uint32_t* output);
icicle_error_t icicle_hip_goldilocks_extension_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u64_t* config, uint32_t* output);
icicle_error_t icicle_hip_goldilocks_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config);
icicle_error_t icicle_hip_goldilocks_ntt_release_domain(void);
icicle_error_t icicle_hip_goldilocks_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou);
ICICLE_HIP_DECLARE_CONVERT(goldilocks)
ICICLE_HIP_DECLARE_CONVERT(icicle_hip_goldilocks)
icicle_error_t goldilocks_extension_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
icicle_error_t icicle_hip_goldilocks_extension_scalar_convert_montgomery(const void* input, uint64_t size, bool is_to_montgomery, const icicle_vec_ops_config_t* config, void* output);
ICICLE_HIP_DECLARE_VEC_ARITH(goldilocks)

#ifdef __cplusplus
}
#endif
#endif /* ICICLE_HIP_H */
