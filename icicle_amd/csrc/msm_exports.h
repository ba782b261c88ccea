// C-ABI export macros of the MSM translation units (msm.hip, msm_curves2.hip, msm_g2.hip, msm_g2_curves2.hip): the
// curves are spread over several files so that they compile in parallel.
#pragma once

// exceptions must never cross the C boundary (Rust/Go callers are extern "C" frames)
#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

// G: bn254 -> group bn254_g1, symbols bn254_msm ...; G2: bn254 -> group bn254_g2, symbols bn254_g2_msm ...
#define DEFINE_MSM_EXPORTS(SYM, GROUP)                                                                                 \
  extern "C" icicle_error_t SYM##_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results) \
  {                                                                                                                    \
    GUARDED(msm_run<GROUP>(scalars, bases, msm_size, config, results));                                                \
  }                                                                                                                    \
  extern "C" icicle_error_t SYM##_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases) \
  {                                                                                                                    \
    GUARDED(msm_precompute_run<GROUP>(input_bases, nof_bases, config, output_bases));                                  \
  }                                                                                                                    \
  extern "C" icicle_error_t icicle_hip_##SYM##_msm(const void* s, const void* b, int n, const icicle_msm_config_t* c, void* r) { GUARDED(msm_run<GROUP>(s, b, n, c, r)); } \
  extern "C" icicle_error_t icicle_hip_##SYM##_msm_precompute_bases(const void* i, int n, const icicle_msm_config_t* c, void* o) { GUARDED(msm_precompute_run<GROUP>(i, n, c, o)); } \
  extern "C" icicle_error_t SYM##_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream)  \
  {                                                                                                                    \
    GUARDED(proj_sum_run<GROUP>(points, n, out, (hipStream_t)stream));                                                 \
  }                                                                                                                    \
  extern "C" icicle_error_t SYM##_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream) \
  {                                                                                                                    \
    GUARDED(generate_run<GROUP>(out, n, k0, out_on_device, (hipStream_t)stream));                                      \
  }
#define DEFINE_G1(C) DEFINE_MSM_EXPORTS(C, C##_g1)
#define DEFINE_G2(C) DEFINE_MSM_EXPORTS(C##_g2, C##_g2)
