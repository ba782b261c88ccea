// Montgomery-form conversion of scalars and points on device -- SURVEY.md 8(f) rank 1: the Rust MSM
// test converts its scalars to Montgomery form ON THE MAIN DEVICE before calling msm
// (wrappers/rust/icicle-core/src/msm/tests.rs:54-59), so a backend without this vec-op cannot run the
// wrapper suite. Reference semantics: x -> x*R or x*R^-1 mod p with R = 2^(32*limbs), element-wise
// (icicle/include/icicle/math/modular_arithmetic.h:583-585; CPU backend:
// icicle/backend/cpu/src/field/cpu_vec_ops.cpp convert_montgomery, backend/cpu/src/curve/cpu_mont_conversion.cpp:12-27;
// C ABI icicle/src/vec_ops.cpp:402-415, icicle/src/curves/montgomery_conversion.cpp:10-58).
// One field multiplication per element: HBM-bound (read + write of the data).
#include "common.h"
#include "bigfield.hpp"
#include "smallfield.hpp"
#include "goldfield.hpp"

namespace icicle_hip {

  // n field elements of PR::NL32 words each
  template <class PR>
  __global__ __launch_bounds__(256) void k_convert_big(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, bool to_mont)
  {
    using F = FieldOps<PR>;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t w[F::N32];
#pragma unroll
    for (int i = 0; i < F::N32; i++)
      w[i] = in[t * F::N32 + i];
    typename F::fe c;
#pragma unroll
    for (int i = 0; i < F::N; i++)
      c.l[i] = to_mont ? PR::CANON_TO_REFMONT[i] : PR::REFMONT_TO_CANON[i];
    BF_SET_BOUND(c, 1);
    F::pack(w, F::reduce(F::mul(F::unpack(w), c)));
#pragma unroll
    for (int i = 0; i < F::N32; i++)
      out[t * F::N32 + i] = w[i];
  }

  // goldilocks: x -> x * 2^64 or x * 2^-64 mod p (goldilocks.h:179-184), n elements of 2 words
  __global__ __launch_bounds__(256) void k_convert_gold(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, bool to_mont)
  {
    using F = FieldOps<goldilocks_params>;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
      const uint2 v = ((const uint2*)in)[t];
      const uint32_t w[2] = {v.x, v.y};
      const GoldFe r = F::mul(F::unpack(w), F::make(to_mont ? goldilocks_params::EPS : goldilocks_params::R_INV));
      ((uint2*)out)[t] = make_uint2((uint32_t)r.v, (uint32_t)(r.v >> 32));
    }
  }

  template <class PR>
  __global__ __launch_bounds__(256) void k_convert_small(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, bool to_mont)
  {
    using S = SmallField<PR>;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride)
      out[t] = to_mont ? S::to_mont(in[t]) : S::from_mont(in[t]);
  }

  // nelem = field elements in total (size * batch * coordinates/lanes), words = u32 per field element
  template <class LAUNCH>
  static icicle_error_t convert_run(const void* input, uint64_t nelem, int words, const icicle_vec_ops_config_t* cfg, void* output, LAUNCH launch)
  {
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (nelem == 0) return ICICLE_SUCCESS;
    if (!input || !output) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const size_t bytes = (size_t)nelem * words * 4;
    TempBuf d_in_tmp, d_out_tmp;
    const uint32_t* d_in = (const uint32_t*)input;
    uint32_t* d_out = (uint32_t*)output;
    if (!cfg->is_a_on_device) {
      HIP_TRY(d_in_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_in_tmp.ptr(), input, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_in = d_in_tmp.as<uint32_t>();
    }
    if (!cfg->is_result_on_device) {
      if (!cfg->is_a_on_device) {
        d_out = d_in_tmp.as<uint32_t>(); // in place in the staging buffer
      } else {
        HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
        d_out = d_out_tmp.as<uint32_t>();
      }
    }
    launch(d_in, d_out, (size_t)nelem, st);
    LAUNCH_CHECK("k_convert", st);
    if (!cfg->is_result_on_device) {
      HIP_TRY(hipMemcpyAsync(output, d_out, bytes, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    } else if (!cfg->is_async) {
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    }
    return ICICLE_SUCCESS;
  }

  template <class PR>
  static icicle_error_t convert_big(const void* in, uint64_t nelem, bool to_mont, const icicle_vec_ops_config_t* cfg, void* out)
  {
    return convert_run(in, nelem, PR::NL32, cfg, out, [&](const uint32_t* i, uint32_t* o, size_t n, hipStream_t st) {
      k_convert_big<PR><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(i, o, n, to_mont);
    });
  }
  template <class PR>
  static icicle_error_t convert_small(const void* in, uint64_t nelem, bool to_mont, const icicle_vec_ops_config_t* cfg, void* out)
  {
    return convert_run(in, nelem, 1, cfg, out, [&](const uint32_t* i, uint32_t* o, size_t n, hipStream_t st) {
      k_convert_small<PR><<<(unsigned)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(i, o, n, to_mont);
    });
  }
  static icicle_error_t convert_gold(const void* in, uint64_t nelem, bool to_mont, const icicle_vec_ops_config_t* cfg, void* out)
  {
    return convert_run(in, nelem, 2, cfg, out, [&](const uint32_t* i, uint32_t* o, size_t n, hipStream_t st) {
      k_convert_gold<<<(unsigned)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(i, o, n, to_mont);
    });
  }
  static uint64_t batch_of(const icicle_vec_ops_config_t* c) { return (c && c->batch_size > 1) ? (uint64_t)c->batch_size : 1; }

} // namespace icicle_hip

using namespace icicle_hip;

#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

// scalar_convert_montgomery: `size` elements per batch entry (icicle/src/vec_ops.cpp:404-408)
#define DEFINE_SCALAR_CONVERT_BIG(NAME, PR)                                                                            \
  extern "C" icicle_error_t NAME##_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, n * batch_of(c), to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, n * batch_of(c), to, c, o)); }
#define DEFINE_SCALAR_CONVERT_SMALL(NAME, PR)                                                                          \
  extern "C" icicle_error_t NAME##_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_small<PR>(i, n * batch_of(c), to, c, o)); } \
  extern "C" icicle_error_t NAME##_extension_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_small<PR>(i, 4 * n * batch_of(c), to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_small<PR>(i, n * batch_of(c), to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_extension_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_small<PR>(i, 4 * n * batch_of(c), to, c, o)); }
// points: n points of 2 (affine) or 3 (projective) base-field coordinates (icicle/src/curves/montgomery_conversion.cpp:12-16,46-50)
#define DEFINE_POINT_CONVERT(NAME, PR)                                                                                 \
  extern "C" icicle_error_t NAME##_affine_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 2 * n, to, c, o)); } \
  extern "C" icicle_error_t NAME##_projective_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 3 * n, to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_affine_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 2 * n, to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_projective_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 3 * n, to, c, o)); }

// G2 points: coordinates are Fq2 = two base-field components each, converted component-wise
// (fields/complex_extension.h:88-96; src/curves/montgomery_conversion.cpp:26-44,60-78)
#define DEFINE_G2_POINT_CONVERT(NAME, PR)                                                                              \
  extern "C" icicle_error_t NAME##_g2_affine_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 4 * n, to, c, o)); } \
  extern "C" icicle_error_t NAME##_g2_projective_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 6 * n, to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_g2_affine_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 4 * n, to, c, o)); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_g2_projective_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_big<PR>(i, 6 * n, to, c, o)); }

DEFINE_SCALAR_CONVERT_BIG(bn254, bn254_fr_params)
DEFINE_SCALAR_CONVERT_BIG(bls12_381, bls12_381_fr_params)
DEFINE_SCALAR_CONVERT_BIG(bls12_377, bls12_377_fr_params)
DEFINE_SCALAR_CONVERT_BIG(grumpkin, bn254_fq_params) // Grumpkin's scalar field is BN254's base field
DEFINE_SCALAR_CONVERT_BIG(stark252, stark252_fr_params)
DEFINE_SCALAR_CONVERT_SMALL(babybear, babybear_params)
DEFINE_SCALAR_CONVERT_SMALL(koalabear, koalabear_params)
DEFINE_POINT_CONVERT(bn254, bn254_fq_params)
DEFINE_POINT_CONVERT(bls12_381, bls12_381_fq_params)
DEFINE_POINT_CONVERT(bls12_377, bls12_377_fq_params)
DEFINE_POINT_CONVERT(grumpkin, bn254_fr_params)
DEFINE_G2_POINT_CONVERT(bn254, bn254_fq_params)
DEFINE_G2_POINT_CONVERT(bls12_381, bls12_381_fq_params)
DEFINE_G2_POINT_CONVERT(bls12_377, bls12_377_fq_params)
// goldilocks and its quadratic extension (two components per element)
extern "C" icicle_error_t goldilocks_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_gold(i, n * batch_of(c), to, c, o)); }
extern "C" icicle_error_t goldilocks_extension_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_gold(i, 2 * n * batch_of(c), to, c, o)); }
extern "C" icicle_error_t icicle_hip_goldilocks_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_gold(i, n * batch_of(c), to, c, o)); }
extern "C" icicle_error_t icicle_hip_goldilocks_extension_scalar_convert_montgomery(const void* i, uint64_t n, bool to, const icicle_vec_ops_config_t* c, void* o) { GUARDED(convert_gold(i, 2 * n * batch_of(c), to, c, o)); }
