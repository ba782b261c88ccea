// G2 instantiation of the MSM for BLS12-377: the twist over Fq2 = Fq[u]/(u^2 + 5) (fq2.hpp NONRES = 5;
// icicle/include/icicle/curves/params/bls12_377.h G2 block).
#include "msm_impl.hpp"

using namespace icicle_hip;

#include "msm_exports.h"
DEFINE_G2(bls12_377)
