// G1 instantiation of the MSM (msm_impl.hpp) and its C ABI. G2 lives in msm_g2.hip so that the two
// halves compile in parallel.
#include "msm_impl.hpp"

using namespace icicle_hip;

// exceptions must never cross the C boundary (Rust/Go callers are extern "C" frames)
#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

extern "C" {
icicle_error_t bn254_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results)
{
  GUARDED(msm_run<bn254_g1>(scalars, bases, msm_size, config, results));
}
icicle_error_t bn254_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases)
{
  GUARDED(msm_precompute_run<bn254_g1>(input_bases, nof_bases, config, output_bases));
}
icicle_error_t bls12_381_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results)
{
  GUARDED(msm_run<bls12_381_g1>(scalars, bases, msm_size, config, results));
}
icicle_error_t bls12_381_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases)
{
  GUARDED(msm_precompute_run<bls12_381_g1>(input_bases, nof_bases, config, output_bases));
}
// Collision-free aliases for the reference-runtime plugin (plugin/): inside a process that also loads
// the reference's libicicle_curve_<c>.so, the plain names above belong to the reference frontend.
icicle_error_t icicle_hip_bn254_msm(const void* s, const void* b, int n, const icicle_msm_config_t* c, void* r) { GUARDED(msm_run<bn254_g1>(s, b, n, c, r)); }
icicle_error_t icicle_hip_bn254_msm_precompute_bases(const void* i, int n, const icicle_msm_config_t* c, void* o) { GUARDED(msm_precompute_run<bn254_g1>(i, n, c, o)); }
icicle_error_t icicle_hip_bls12_381_msm(const void* s, const void* b, int n, const icicle_msm_config_t* c, void* r) { GUARDED(msm_run<bls12_381_g1>(s, b, n, c, r)); }
icicle_error_t icicle_hip_bls12_381_msm_precompute_bases(const void* i, int n, const icicle_msm_config_t* c, void* o) { GUARDED(msm_precompute_run<bls12_381_g1>(i, n, c, o)); }
icicle_error_t bn254_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream)
{
  GUARDED(proj_sum_run<bn254_g1>(points, n, out, (hipStream_t)stream));
}
icicle_error_t bls12_381_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream)
{
  GUARDED(proj_sum_run<bls12_381_g1>(points, n, out, (hipStream_t)stream));
}
icicle_error_t bn254_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream)
{
  GUARDED(generate_run<bn254_g1>(out, n, k0, out_on_device, (hipStream_t)stream));
}
icicle_error_t bls12_381_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream)
{
  GUARDED(generate_run<bls12_381_g1>(out, n, k0, out_on_device, (hipStream_t)stream));
}
// the window plan the backend would use (bench.py's operation counts)
icicle_error_t icicle_hip_msm_plan(int msm_size, int scalar_bits, const icicle_msm_config_t* config, int* c, int* nwin)
{
  if (!config || !c || !nwin || msm_size < 0 || scalar_bits <= 0) return ICICLE_INVALID_ARGUMENT;
  const MsmPlan pl = make_plan(std::max(msm_size, 1), scalar_bits, *config);
  *c = pl.c;
  *nwin = pl.nwin;
  return ICICLE_SUCCESS;
}
// gives the per-device copies made under config.ext "hip_bases_resident" back (bases == NULL: all of them)
icicle_error_t icicle_hip_msm_release_resident_bases(const void* bases)
{
  try {
    (void)resident_release(bases);
    return ICICLE_SUCCESS;
  } catch (...) {
    return ICICLE_INVALID_ARGUMENT;
  }
}
}
