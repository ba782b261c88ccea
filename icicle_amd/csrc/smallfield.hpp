// 31-bit prime fields (BabyBear p = 2^31 - 2^27 + 1, KoalaBear p = 2^31 - 2^24 + 1) for the NTT.
// The reference multiplies then Barrett-reduces (modular_arithmetic.h:517-521, host_math.h:438-470);
// on gfx950 the cheapest exact form is single-word Montgomery with R = 2^32: one v_mad_u64_u32
// (or mul_lo+mul_hi) for the product, one v_mul_lo_u32 and one v_mul_hi_u32 for the reduction.
// Elements are kept in Montgomery form inside kernels and twiddle tables; I/O is the reference's
// canonical residues in [0,p).
#pragma once
#include <cstdint>
#include "field_consts.h"

#if defined(__HIPCC__)
  #include <hip/hip_runtime.h>
  #define SF_HD __host__ __device__ __forceinline__
#else
  #define SF_HD inline __attribute__((always_inline))
#endif

namespace icicle_hip {

  template <class PR>
  struct SmallField {
    static constexpr uint32_t P = PR::P;

    static SF_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
    static SF_HD uint32_t add(uint32_t a, uint32_t b)
    {
      uint32_t s = a + b;
#if defined(NTT_EXP_NO_CORRECTION) // WHAT-IF build only (wrong results): how much of a pass is the add/sub correction work?
      return s;
#else
      return umin(s, s - P);
#endif
    }
    static SF_HD uint32_t sub(uint32_t a, uint32_t b)
    {
      uint32_t d = a - b;
#if defined(NTT_EXP_NO_CORRECTION)
      return d;
#else
      return umin(d, d + P);
#endif
    }
    static SF_HD uint32_t neg(uint32_t a) { return sub(0, a); }
    // t < p * 2^32  ->  t / 2^32 mod p, in [0,p)
    static SF_HD uint32_t mont_reduce(uint64_t t)
    {
      // t + m'*p in ONE v_mad_u64_u32 with m' = -t/p mod 2^32: the sum is divisible by 2^32 and below 2p*2^32 (no
      // carry out of 64 bits since t, m'*p < p*2^32), so its high word is the result in [0, 2p) -- 5 instructions per
      // product instead of 6 for the subtract-the-high-words form (64 x 2^24 BabyBear: 6.06 -> 5.98 ms, same box)
      uint32_t m = (uint32_t)t * (0u - PR::PINV);
      uint32_t r = (uint32_t)((t + (uint64_t)m * P) >> 32);
      return umin(r, r - P);
    }
    static SF_HD uint32_t mul(uint32_t a, uint32_t b) { return mont_reduce((uint64_t)a * b); }
    static SF_HD uint32_t to_mont(uint32_t x) { return mul(x, PR::R2); }
    static SF_HD uint32_t from_mont(uint32_t x) { return mont_reduce((uint64_t)x); }
    static SF_HD uint32_t one() { return PR::ONE; }
    static SF_HD uint32_t pow(uint32_t x, uint64_t e)
    { // x Montgomery, e plain
      uint32_t r = one();
      while (e) {
        if (e & 1) r = mul(r, x);
        x = mul(x, x);
        e >>= 1;
      }
      return r;
    }
    static SF_HD uint32_t inv(uint32_t x) { return pow(x, (uint64_t)P - 2); }
  };

} // namespace icicle_hip
