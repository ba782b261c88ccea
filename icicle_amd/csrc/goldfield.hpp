// Goldilocks, p = 2^64 - 2^32 + 1 (reference: icicle/include/icicle/fields/stark_fields/goldilocks.h:206-276, FIELD_ID
// 1005): FieldOps<goldilocks_params>, a specialisation with the static interface the NTT over multi-word elements
// (ntt_big.hip), the vector ops and the Montgomery conversion use -- so the field rides on the kernels of the 256-bit
// scalar fields with W = 2 words per element.
//
// Elements are kept CANONICAL in one 64-bit register: the special form of p makes a plain product cheap to reduce
// (2^64 = 2^32 - 1 and 2^96 = -1 mod p: two 64-bit additions with carry fix-ups after the 64 x 64 -> 128 product), so
// there is no Montgomery form on the device -- "unpack" / "from_canonical" are the identity and the twiddle tables hold
// canonical values. The reference's Montgomery form (x * 2^64 mod p, goldilocks.h:179-184, montgomery_r = 2^32 - 1) exists
// at the boundary only (scalar_convert_montgomery). Every operation returns a value in [0, p): the lazy-bound template
// arguments of the shared kernels (sub<K>) mean nothing here.
#pragma once
#include "bigfield.hpp"

namespace icicle_hip {

  struct goldilocks_params {
    static constexpr int NL = 2;   // (unused: no 29-bit limbs)
    static constexpr int NL32 = 2; // packed 32-bit words (reference storage<2>)
    static constexpr int NBITS = 64;
    static constexpr uint64_t P = 0xFFFFFFFF00000001ull;
    static constexpr uint64_t EPS = 0xFFFFFFFFull;              // 2^64 mod p = 2^32 - 1 = the reference's montgomery_r
    static constexpr uint64_t R_INV = 0xFFFFFFFE00000001ull;    // (2^64)^-1 mod p = montgomery_r_inv (goldilocks.h:225)
    static constexpr int TWO_ADICITY = 32;                      // omegas_count (goldilocks.h:257)
    // word views padded to eight words: the host helpers shared with the 256-bit fields walk eight words of a modulus
    static constexpr uint32_t P32[8] = {0x00000001u, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0};
    static constexpr uint32_t ROU32[8] = {0xda58878cu, 0x185629dcu, 0, 0, 0, 0, 0, 0}; // order 2^32 (goldilocks.h:256)
    static constexpr uint32_t EXT_NONRES = 7; // quadratic extension u^2 = 7 (goldilocks.h:268-270), NTT is lane-wise
  };

  struct GoldFe {
    uint64_t v;
  };

  template <>
  struct FieldOps<goldilocks_params> {
    using PR = goldilocks_params;
    static constexpr int N = 2;
    static constexpr int N32 = 2;
    static constexpr bool TIGHT = false;
    using fe = GoldFe;
    static constexpr uint64_t P = PR::P, EPS = PR::EPS;

    static HD fe make(uint64_t v)
    {
      fe r;
      r.v = v;
      return r;
    }
    static HD fe zero() { return make(0); }
    static HD fe one() { return make(1); }

    static HD fe add(const fe& a, const fe& b)
    {
      uint64_t s = a.v + b.v;
      if (s < a.v)
        s += EPS; // wrapped: + 2^64 = + EPS (mod p); a + b < 2p keeps this below p
      else if (s >= P)
        s -= P;
      return make(s);
    }
    static HD fe dbl(const fe& a) { return add(a, a); }
    template <int K>
    static HD fe sub(const fe& a, const fe& b)
    {
      uint64_t d = a.v - b.v;
      if (a.v < b.v) d -= EPS; // wrapped: - 2^64 + p = - EPS
      return make(d);
    }
    template <int K>
    static HD fe neg(const fe& a)
    {
      return make(a.v ? P - a.v : 0);
    }
    // (hi : lo) mod p, canonical
    static HD uint64_t reduce128(uint64_t hi, uint64_t lo)
    {
      const uint64_t hh = hi >> 32, hl = hi & EPS;
      uint64_t t0 = lo - hh; // 2^96 = -1
      if (lo < hh) t0 -= EPS;
      const uint64_t t1 = hl * EPS; // 2^64 = 2^32 - 1
      uint64_t r = t0 + t1;
      if (r < t1) r += EPS;
      if (r >= P) r -= P;
      return r;
    }
    static HD fe mul(const fe& a, const fe& b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
      return make(reduce128(__umul64hi(a.v, b.v), a.v * b.v));
#else
      const unsigned __int128 m = (unsigned __int128)a.v * b.v;
      return make(reduce128((uint64_t)(m >> 64), (uint64_t)m));
#endif
    }
    static HD fe sqr(const fe& a) { return mul(a, a); }
    static HD fe select(bool c, const fe& a, const fe& b) { return c ? a : b; }
    static HD fe reduce(const fe& a) { return a; }
    static HD bool is_zero(const fe& a) { return a.v == 0; }
    static HD bool eq(const fe& a, const fe& b) { return a.v == b.v; }
    static HD fe inv(const fe& a)
    { // a^(p-2); 0 -> 0
      fe r = one(), base = a;
      const uint64_t e = P - 2;
      for (int b = 0; b < 64; b++) {
        if ((e >> b) & 1) r = mul(r, base);
        base = sqr(base);
      }
      return r;
    }

    // words <-> element (canonical both ways; an input in [p, 2^64) is brought below p)
    static HD fe unpack(const uint32_t* w)
    {
      uint64_t v = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
      if (v >= P) v -= P;
      return make(v);
    }
    static HD void pack(uint32_t* w, const fe& a)
    {
      w[0] = (uint32_t)a.v;
      w[1] = (uint32_t)(a.v >> 32);
    }
    static HD fe from_canonical(const uint32_t* w) { return unpack(w); }
    static HD void to_canonical(uint32_t* w, const fe& a) { pack(w, a); }
    static HD fe from_refmont(const uint32_t* w) { return mul(unpack(w), make(PR::R_INV)); }
    static HD void to_refmont(uint32_t* w, const fe& a) { pack(w, mul(a, make(EPS))); }
  };

} // namespace icicle_hip
