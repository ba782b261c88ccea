// G2 instantiation of the MSM (msm_impl.hpp over fq2.hpp) and its C ABI:
// <curve>_g2_msm / _g2_msm_precompute_bases (icicle/src/msm.cpp:25-41,58-74, built with G2_ENABLED).
// g2_affine_t = {x.c0, x.c1, y.c0, y.c1}, g2_projective_t adds z (curves/params/bn254.h:15-17).
#include "msm_impl.hpp"

using namespace icicle_hip;

#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

#define DEFINE_G2(C)                                                                                                   \
  extern "C" icicle_error_t C##_g2_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results) \
  {                                                                                                                    \
    GUARDED(msm_run<C##_g2>(scalars, bases, msm_size, config, results));                                               \
  }                                                                                                                    \
  extern "C" icicle_error_t C##_g2_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases) \
  {                                                                                                                    \
    GUARDED(msm_precompute_run<C##_g2>(input_bases, nof_bases, config, output_bases));                                 \
  }                                                                                                                    \
  extern "C" icicle_error_t icicle_hip_##C##_g2_msm(const void* s, const void* b, int n, const icicle_msm_config_t* c, void* r) { GUARDED(msm_run<C##_g2>(s, b, n, c, r)); } \
  extern "C" icicle_error_t icicle_hip_##C##_g2_msm_precompute_bases(const void* i, int n, const icicle_msm_config_t* c, void* o) { GUARDED(msm_precompute_run<C##_g2>(i, n, c, o)); } \
  extern "C" icicle_error_t C##_g2_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream) \
  {                                                                                                                    \
    GUARDED(proj_sum_run<C##_g2>(points, n, out, (hipStream_t)stream));                                                \
  }                                                                                                                    \
  extern "C" icicle_error_t C##_g2_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream) \
  {                                                                                                                    \
    GUARDED(generate_run<C##_g2>(out, n, k0, out_on_device, (hipStream_t)stream));                                     \
  }

DEFINE_G2(bn254)
DEFINE_G2(bls12_381)
