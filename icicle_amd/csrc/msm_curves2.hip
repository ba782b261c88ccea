// G1 instantiation of the MSM (msm_impl.hpp) for the further curves of the reference that have an MSM
// (icicle/cmake/features.cmake:17,19): BLS12-377 and Grumpkin. bw6_761 (761-bit base field) is not built.
#include "msm_impl.hpp"

using namespace icicle_hip;

#include "msm_exports.h"
DEFINE_G1(bls12_377)
DEFINE_G1(grumpkin)
