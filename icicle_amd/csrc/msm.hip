// G1 instantiation of the MSM (msm_impl.hpp) and its C ABI. G2 lives in msm_g2.hip so that the two
// halves compile in parallel.
#include "msm_impl.hpp"

using namespace icicle_hip;

// <curve>_msm / _msm_precompute_bases (icicle/src/msm.cpp:12-16,45-49) + collision-free aliases for the reference-runtime
// plugin (plugin/): inside a process that also loads the reference's libicicle_curve_<c>.so, the plain names belong to the
// reference frontend. Shared with msm_curves2.hip.
#include "msm_exports.h"
DEFINE_G1(bn254)
DEFINE_G1(bls12_381)

extern "C" {
// the window plan the backend would use (bench.py's operation counts)
icicle_error_t icicle_hip_msm_plan(int msm_size, int scalar_bits, const icicle_msm_config_t* config, int* c, int* nwin)
{
  if (!config || !c || !nwin || msm_size < 0 || scalar_bits <= 0) return ICICLE_INVALID_ARGUMENT;
  const MsmPlan pl = make_plan(std::max(msm_size, 1), scalar_bits, *config);
  *c = pl.c;
  *nwin = pl.nwin;
  return ICICLE_SUCCESS;
}
// the whole plan: out[0..7] = c (bits of the widest windows), nwin, n_lo (windows one bit narrower: mixed-width plans), negate
// (negate-if-top-bit, cpu_msm.hpp:276-277), buckets per window slot, buckets in use, segment size, windows per base-table entry.
// force_windows as MSMConfig.ext "hip_msm_windows" (0 = the cost model decides, as msm() does).
icicle_error_t icicle_hip_msm_plan_info(int msm_size, int scalar_bits, const icicle_msm_config_t* config, int force_windows, int* out)
{
  if (!config || !out || msm_size < 0 || scalar_bits <= 0) return ICICLE_INVALID_ARGUMENT;
  const MsmPlan pl = make_plan(std::max(msm_size, 1), scalar_bits, *config, force_windows);
  out[0] = pl.c, out[1] = pl.nwin, out[2] = pl.n_lo, out[3] = pl.negate ? 1 : 0;
  out[4] = (int)pl.nb, out[5] = (int)std::min<size_t>(pl.buckets_used(), 0x7fffffff), out[6] = (int)pl.seg, out[7] = pl.wpf;
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
