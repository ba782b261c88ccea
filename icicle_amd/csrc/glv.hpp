// GLV decomposition of a scalar for the j = 0 curves of this backend (BN254, BLS12-381, BLS12-377, Grumpkin).
//
// phi(x, y) = (beta x, y) is an endomorphism that acts on the r-torsion as multiplication by lambda (beta^3 = 1 in Fq,
// lambda^2 + lambda + 1 = 0 mod r), so k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^129: one joint chain of 128 doublings instead of
// 252. Used by the ECNTT's butterflies (ecntt.hip), whose cost IS the length of that chain; the MSM gains nothing from it (the
// bucket method pays per point and window, and 2 n half-length scalars cost what n full-length ones do).
// The reference multiplies with a plain windowed double-and-add (icicle/include/icicle/curves/projective.h:192-224); any method
// that returns the same group element is a drop-in.
// Precondition: P in the subgroup of prime order r (phi = lambda only there; BLS12-381 / BLS12-377 have cofactors of ~2^126). The ECNTT
// itself has that precondition: outside the subgroup w^a (w^b P) != w^(a + b mod r) P and the reference's own transform is neither
// invertible nor equal to its definition (tests/test_ecntt_domain_of_definition.py pins that with the reference), so the split
// narrows nothing.
//
// Constants (tools/gen_consts.py glv_constants -> field_consts.h): a reduced basis (a1, b1), (a2, b2) of the lattice
// {(a, b) : a + b lambda = 0 mod r} and G_i = 2^256 b2 / det, 2^256 (-b1) / det rounded towards zero. With
//   c_i = sign(G_i) ((k |G_i|) >> 256),   k1 = k - c1 a1 - c2 a2,   k2 = -c1 b1 - c2 b2
// k1 + k2 lambda = k (mod r) holds EXACTLY for any integers c_i (the basis vectors are in the lattice); the rounding only bounds the
// size. All arithmetic below is two's complement mod 2^160 -- the true values fit 130 bits -- with static word indices.
// tests/test_host_math.py checks this very code (host build) against Python integers.
#pragma once
#include <cstdint>
#include "field_consts.h"

#if defined(__HIPCC__)
  #include <hip/hip_runtime.h>
  #define GLV_HD __host__ __device__ __forceinline__
#else
  #define GLV_HD inline __attribute__((always_inline))
#endif

namespace icicle_hip {

  // r[0..4] = words 8..12 of a[0..7] * b[0..4]  (= (a * b) >> 256 for a < 2^256, b < 2^160: the product has 13 words)
  GLV_HD void glv_mul_hi(uint32_t* r, const uint32_t* a, const uint32_t* b)
  {
    uint32_t w[13];
#pragma unroll
    for (int i = 0; i < 13; i++)
      w[i] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      uint64_t carry = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint64_t t = (uint64_t)a[i] * b[j] + w[i + j] + carry;
        w[i + j] = (uint32_t)t;
        carry = t >> 32;
      }
      w[j + 8] = (uint32_t)carry;
    }
#pragma unroll
    for (int i = 0; i < 5; i++)
      r[i] = w[8 + i];
  }
  // r = a * b mod 2^160 (5 words each; two's complement operands give the two's complement product)
  GLV_HD void glv_mul_lo(uint32_t* r, const uint32_t* a, const uint32_t* b)
  {
#pragma unroll
    for (int i = 0; i < 5; i++)
      r[i] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      uint64_t carry = 0;
#pragma unroll
      for (int i = 0; i + j < 5; i++) {
        const uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + carry;
        r[i + j] = (uint32_t)t;
        carry = t >> 32;
      }
    }
  }
  GLV_HD void glv_neg(uint32_t* a) // a = -a mod 2^160
  {
    uint64_t c = 1;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint64_t t = (uint64_t)(~a[i]) + c;
      a[i] = (uint32_t)t;
      c = t >> 32;
    }
  }
  GLV_HD void glv_sub(uint32_t* a, const uint32_t* b) // a -= b mod 2^160
  {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint64_t t = (uint64_t)a[i] - b[i] - borrow;
      a[i] = (uint32_t)t;
      borrow = (t >> 32) & 1u;
    }
  }

  // k (8 canonical words, < r)  ->  |k1|, |k2| (5 words each, < 2^130) and their signs
  template <class C>
  GLV_HD void glv_decompose(const uint32_t* k, uint32_t* k1, bool& neg1, uint32_t* k2, bool& neg2)
  {
    uint32_t c1[5], c2[5], t[5];
    {
      uint32_t g[5];
#pragma unroll
      for (int i = 0; i < 5; i++)
        g[i] = C::GLV_G1[i];
      glv_mul_hi(c1, k, g);
      if (C::GLV_G1_NEG) glv_neg(c1);
#pragma unroll
      for (int i = 0; i < 5; i++)
        g[i] = C::GLV_G2[i];
      glv_mul_hi(c2, k, g);
      if (C::GLV_G2_NEG) glv_neg(c2);
    }
    uint32_t m[5];
#pragma unroll
    for (int i = 0; i < 5; i++)
      k1[i] = k[i], k2[i] = 0; // (k mod 2^160: the result is taken mod 2^160 anyway)
#pragma unroll
    for (int i = 0; i < 5; i++)
      m[i] = C::GLV_A1[i];
    glv_mul_lo(t, c1, m);
    glv_sub(k1, t);
#pragma unroll
    for (int i = 0; i < 5; i++)
      m[i] = C::GLV_A2[i];
    glv_mul_lo(t, c2, m);
    glv_sub(k1, t);
#pragma unroll
    for (int i = 0; i < 5; i++)
      m[i] = C::GLV_B1[i];
    glv_mul_lo(t, c1, m);
    glv_sub(k2, t);
#pragma unroll
    for (int i = 0; i < 5; i++)
      m[i] = C::GLV_B2[i];
    glv_mul_lo(t, c2, m);
    glv_sub(k2, t);
    neg1 = (k1[4] >> 31) != 0;
    neg2 = (k2[4] >> 31) != 0;
    if (neg1) glv_neg(k1);
    if (neg2) glv_neg(k2);
  }

  // |k| < 2^130 (5 words) -> 27 signed five-bit digits, least significant first: k = sum_i d_i 32^i, d_i in [-15, 16]. One digit per
  // byte of pk[0..6]: bits 0..4 = |d_i| (0..16), bit 7 = sign. A window value above 16 becomes value - 32 with a carry into the next
  // window, so a table of the multiples 1..16 serves (with one conditional negation) where unsigned five-bit windows would need 31.
  GLV_HD void glv_recode5(const uint32_t* k, uint32_t* pk)
  {
#pragma unroll
    for (int i = 0; i < 7; i++)
      pk[i] = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 27; i++) {
      const int b = 5 * i, w = b >> 5, sh = b & 31;
      uint32_t v = k[w] >> sh;
      if (sh > 27 && w + 1 < 5) v |= k[w + 1] << (32 - sh);
      v = (v & 31u) + carry;
      const bool negd = v > 16u;
      carry = negd ? 1u : 0u;
      const uint32_t mag = negd ? 32u - v : v;
      pk[i >> 2] |= (mag | (negd ? 0x80u : 0u)) << ((i & 3) * 8);
    }
  }

} // namespace icicle_hip
