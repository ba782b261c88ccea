// G2 instantiation of the MSM (msm_impl.hpp over fq2.hpp) and its C ABI:
// <curve>_g2_msm / _g2_msm_precompute_bases (icicle/src/msm.cpp:25-41,58-74, built with G2_ENABLED).
// g2_affine_t = {x.c0, x.c1, y.c0, y.c1}, g2_projective_t adds z (curves/params/bn254.h:15-17).
#include "msm_impl.hpp"

using namespace icicle_hip;

#include "msm_exports.h"

DEFINE_G2(bn254)
DEFINE_G2(bls12_381)
