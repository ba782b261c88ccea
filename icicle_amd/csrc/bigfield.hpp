// Big prime-field arithmetic for gfx950, written around the one cheap wide integer op CDNA4 has:
// v_mad_u64_u32 (32x32+64 -> 64, ~1.25x the cost of a plain VALU op, measured in
// profiles/r01_alu_ubench.txt). There is NO carry-in on that instruction and carry chains through
// VCC cost a second VALU op per product, so instead of the reference's 32-bit limbs + carry chain
// (icicle/include/icicle/math/host_math.h:227-281, Barrett 438-470) elements are held in
// radix-2^29 limbs: a column of up to 2*NL 58-bit products accumulates in ONE 64-bit register pair
// with no carry handling at all, and Montgomery reduction (R = 2^(29*NL)) is interleaved in the
// same accumulator (product scanning / FIPS).
//
// Value discipline ("lazy reduction"): R/p >= 64 for every field here, so a Montgomery product of
// inputs < Ka*p, Kb*p is < (Ka*Kb/(R/p) + 1)*p. Nothing on the hot path is ever conditionally
// reduced; subtraction adds a multiple of p (K*p, K in {2,4,8,16}) big enough to stay positive.
// A normalised element has limbs 0..NL-2 in [0,2^29) and a non-negative top limb.
// In host debug builds (-DBIGFIELD_BOUNDS) every element carries its bound (in units of p) and each
// op asserts its precondition, so the bounds argument of ec.hpp is machine-checked by tests.
//
// I/O is the reference's canonical storage<N> (N x u32 little-endian, value in [0,p), NOT
// Montgomery unless a *_montgomery_form flag says so; modular_arithmetic.h:517-521, 583-585).
#pragma once
#include <cstdint>
#include "field_consts.h"
#include "mont_asm.hpp" // device builds: the products below as single inline-asm blocks (-DBIGFIELD_NO_ASM turns them off)

#if defined(__HIPCC__)
  #include <hip/hip_runtime.h>
  #define HD __host__ __device__ __forceinline__
#else
  #define HD inline __attribute__((always_inline))
#endif

#ifdef BIGFIELD_BOUNDS
  #include <cassert>
  #include <cstdio>
  #define BF_BOUND_DECL double bnd = 0;
  #define BF_SET_BOUND(x, v) (x).bnd = (v)
  #define BF_ASSERT(cond, msg)                                                                                         \
    do {                                                                                                               \
      if (!(cond)) {                                                                                                   \
        fprintf(stderr, "bigfield bound violation: %s (%s:%d)\n", msg, __FILE__, __LINE__);                            \
        assert(false);                                                                                                 \
      }                                                                                                                \
    } while (0)
#else
  #define BF_BOUND_DECL
  #define BF_SET_BOUND(x, v)
  #define BF_ASSERT(cond, msg)
#endif

namespace icicle_hip {

  template <class PR>
  struct Fe {
    uint32_t l[PR::NL];
    BF_BOUND_DECL
  };

  template <class PR>
  struct FieldOps {
    static constexpr int N = PR::NL;
    static constexpr int N32 = PR::NL32;
    static constexpr uint32_t MASK = RB_MASK;
    static constexpr bool TIGHT = false; // see fq2.hpp
    using fe = Fe<PR>;
    // R/p (lower bound) used only by the debug bound tracker
    static constexpr double r_over_p() { return (double)(1ull << (RB * N - PR::NBITS)); }
    // largest bound (units of p) a stored element may have: value < 2^(RB*N) and mul inputs sane
    static constexpr double max_bound() { return r_over_p() < 64.0 ? r_over_p() : 64.0; }

    static HD fe zero()
    {
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++)
        r.l[i] = 0;
      BF_SET_BOUND(r, 0);
      return r;
    }
    static HD fe one()
    { // Montgomery one
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++)
        r.l[i] = PR::ONE[i];
      BF_SET_BOUND(r, 1);
      return r;
    }
    // R^2 mod p as limbs: mul(x_plain, r2()) = x*R (into Montgomery form); also "1" in the doubly-scaled form x*R^2
    static HD fe r2()
    {
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++)
        r.l[i] = PR::R2[i];
      BF_SET_BOUND(r, 1);
      return r;
    }
    // the plain integer 1: mul(x, plain_one()) = x/R (one Montgomery scaling removed)
    static HD fe plain_one()
    {
      fe r = zero();
      r.l[0] = 1;
      BF_SET_BOUND(r, 1);
      return r;
    }
    using bfe = fe;
    static HD bfe base_r2() { return r2(); }
    static HD bfe base_plain_one() { return plain_one(); }
    static HD fe mul_base(const fe& a, const bfe& s) { return mul(a, s); }
    // constant given as [N] Montgomery limbs (< p)
    template <class ARR>
    static HD fe from_const(const ARR& c)
    {
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++)
        r.l[i] = c[i];
      BF_SET_BOUND(r, 1);
      return r;
    }
    template <int K>
    static HD uint32_t kp(int i)
    {
      static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16, "K");
      if constexpr (K == 1) return PR::P[i];
      if constexpr (K == 2) return PR::P2[i];
      if constexpr (K == 4) return PR::P4[i];
      if constexpr (K == 8) return PR::P8[i];
      return PR::P16[i];
    }

    // ---- Montgomery multiplication: r = a*b/R mod p (lazy), product scanning ----
    // Column k accumulates sum_{i+j=k} a_i*b_j + m_i*p_j in one 64-bit register: <= 2N products of
    // < 2^58 plus a < 2^35 carry; 2N*2^58 < 2^64 for N <= 31.
    static HD fe mul(const fe& a, const fe& b)
    {
      BF_ASSERT(a.bnd <= max_bound() && b.bnd <= max_bound(), "mul input bound");
      fe r;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BIGFIELD_NO_ASM)
      if constexpr (N == 9) {
        mont_mul_asm9<PR>(r.l, a.l, b.l);
        return r;
      } else if constexpr (N == 14) {
        mont_mul_asm14<PR>(r.l, a.l, b.l);
        return r;
      }
#endif
      uint32_t m[N];
      uint64_t acc = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < k; i++) {
          acc += (uint64_t)a.l[i] * b.l[k - i];
          acc += (uint64_t)m[i] * PR::P[k - i];
        }
        acc += (uint64_t)a.l[k] * b.l[0];
        m[k] = ((uint32_t)acc * PR::PINV) & MASK;
        acc += (uint64_t)m[k] * PR::P[0];
        acc >>= RB;
      }
#pragma unroll
      for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
          acc += (uint64_t)a.l[i] * b.l[k - i];
          acc += (uint64_t)m[i] * PR::P[k - i];
        }
        r.l[k - N] = (uint32_t)acc & MASK;
        acc >>= RB;
      }
      r.l[N - 1] = (uint32_t)acc;
      BF_SET_BOUND(r, a.bnd * b.bnd / r_over_p() + 1.0);
      return r;
    }

    // a <- a*b/R (same value and bound as mul). On the device the result is produced in a's own registers, so a
    // loop-carried accumulator coordinate needs no copies at the back edge (mont_asm.hpp).
    static HD void mul_inplace(fe& a, const fe& b)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BIGFIELD_NO_ASM)
      if constexpr (N == 9) {
        mont_mul_inplace_asm9<PR>(a.l, b.l);
        return;
      } else if constexpr (N == 14) {
        mont_mul_inplace_asm14<PR>(a.l, b.l);
        return;
      }
#endif
      a = mul(a, b);
    }

    // r = (a*b + c*d)/R mod p with ONE interleaved reduction: both products share the column
    // accumulators (3N products of < 2^58 per column: < 2^64 for N <= 21).
    static HD fe mul_add(const fe& a, const fe& b, const fe& c, const fe& d)
    {
      BF_ASSERT(a.bnd <= max_bound() && b.bnd <= max_bound() && c.bnd <= max_bound() && d.bnd <= max_bound(), "mul_add input bound");
      static_assert(N <= 21, "column accumulator would overflow");
      fe r;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BIGFIELD_NO_ASM)
      if constexpr (N == 9) {
        mont_mul_add_asm9<PR>(r.l, a.l, b.l, c.l, d.l);
        return r;
      } else if constexpr (N == 14) {
        mont_mul_add_asm14<PR>(r.l, a.l, b.l, c.l, d.l);
        return r;
      }
#endif
      uint32_t m[N];
      uint64_t acc = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < k; i++) {
          acc += (uint64_t)a.l[i] * b.l[k - i];
          acc += (uint64_t)c.l[i] * d.l[k - i];
          acc += (uint64_t)m[i] * PR::P[k - i];
        }
        acc += (uint64_t)a.l[k] * b.l[0];
        acc += (uint64_t)c.l[k] * d.l[0];
        m[k] = ((uint32_t)acc * PR::PINV) & MASK;
        acc += (uint64_t)m[k] * PR::P[0];
        acc >>= RB;
      }
#pragma unroll
      for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
          acc += (uint64_t)a.l[i] * b.l[k - i];
          acc += (uint64_t)c.l[i] * d.l[k - i];
          acc += (uint64_t)m[i] * PR::P[k - i];
        }
        r.l[k - N] = (uint32_t)acc & MASK;
        acc >>= RB;
      }
      r.l[N - 1] = (uint32_t)acc;
      BF_SET_BOUND(r, (a.bnd * b.bnd + c.bnd * d.bnd) / r_over_p() + 1.0);
      return r;
    }

    // c <- (a*b + c*d)/R, result in c's own registers on the device (see mul_inplace)
    static HD void mul_add_inplace_c(fe& c, const fe& a, const fe& b, const fe& d)
    {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BIGFIELD_NO_ASM)
      if constexpr (N == 9) {
        mont_mul_add_inplace_c_asm9<PR>(c.l, a.l, b.l, d.l);
        return;
      } else if constexpr (N == 14) {
        mont_mul_add_inplace_c_asm14<PR>(c.l, a.l, b.l, d.l);
        return;
      }
#endif
      c = mul_add(a, b, c, d);
    }

    // Squaring: the a_i*a_j cross terms are computed once and doubled (N(N+1)/2 instead of N^2
    // products for the a*a half; the m*p half is unchanged).
    static HD fe sqr(const fe& a)
    {
      BF_ASSERT(a.bnd <= max_bound(), "sqr input bound");
      fe r;
      uint32_t m[N];
      uint32_t a2[N]; // 2*a_i (< 2^30)
#pragma unroll
      for (int i = 0; i < N; i++)
        a2[i] = a.l[i] << 1;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BIGFIELD_NO_ASM)
      if constexpr (N == 9) {
        mont_sqr_asm9<PR>(r.l, a.l, a2);
        return r;
      } else if constexpr (N == 14) {
        mont_sqr_asm14<PR>(r.l, a.l, a2);
        return r;
      }
#endif
      uint64_t acc = 0;
#pragma unroll
      for (int k = 0; k < 2 * N - 1; k++) {
        // a*a part of column k: pairs (i, k-i) with i < k-i use the doubled limb, plus the square
        const int lo = (k < N) ? 0 : k - N + 1;
        const int hi = (k < N) ? k : N - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
          const int j = k - i;
          if (i < j) acc += (uint64_t)a2[i] * a.l[j];
          if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
        }
        if (k < N) {
#pragma unroll
          for (int i = 0; i < k; i++)
            acc += (uint64_t)m[i] * PR::P[k - i];
          m[k] = ((uint32_t)acc * PR::PINV) & MASK;
          acc += (uint64_t)m[k] * PR::P[0];
        } else {
#pragma unroll
          for (int i = k - N + 1; i < N; i++)
            acc += (uint64_t)m[i] * PR::P[k - i];
          r.l[k - N] = (uint32_t)acc & MASK;
        }
        acc >>= RB;
      }
      r.l[N - 1] = (uint32_t)acc;
      BF_SET_BOUND(r, a.bnd * a.bnd / r_over_p() + 1.0);
      return r;
    }

    // ---- add / sub with carry normalisation ----
    static HD fe add(const fe& a, const fe& b)
    {
      fe r;
      uint32_t c = 0;
#pragma unroll
      for (int i = 0; i < N - 1; i++) {
        uint32_t t = a.l[i] + b.l[i] + c;
        r.l[i] = t & MASK;
        c = t >> RB;
      }
      r.l[N - 1] = a.l[N - 1] + b.l[N - 1] + c;
      BF_SET_BOUND(r, a.bnd + b.bnd);
      BF_ASSERT(r.bnd <= max_bound(), "add result bound");
      return r;
    }
    static HD fe dbl(const fe& a) { return add(a, a); }

    // r = a - b + K*p ; requires b < K*p (so the value stays non-negative)
    template <int K>
    static HD fe sub(const fe& a, const fe& b)
    {
      BF_ASSERT(b.bnd <= (double)K, "sub: subtrahend exceeds K*p");
      fe r;
      int32_t c = 0;
#pragma unroll
      for (int i = 0; i < N - 1; i++) {
        int32_t t = (int32_t)(a.l[i] - b.l[i] + kp<K>(i)) + c;
        r.l[i] = (uint32_t)t & MASK;
        c = t >> RB; // arithmetic
      }
      r.l[N - 1] = a.l[N - 1] - b.l[N - 1] + kp<K>(N - 1) + (uint32_t)c;
      BF_SET_BOUND(r, a.bnd + (double)K);
      BF_ASSERT(r.bnd <= max_bound(), "sub result bound");
      return r;
    }
    template <int K>
    static HD fe neg(const fe& a)
    {
      return sub<K>(zero(), a);
    }

    // r = cond ? a : b (branch-free select)
    static HD fe select(bool cond, const fe& a, const fe& b)
    {
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++)
        r.l[i] = cond ? a.l[i] : b.l[i];
#ifdef BIGFIELD_BOUNDS
      r.bnd = a.bnd > b.bnd ? a.bnd : b.bnd;
#endif
      return r;
    }

    // ---- full reduction to [0,p) (output path only) ----
    // one conditional subtraction of K*p
    template <int K>
    static HD void cond_sub(fe& a)
    {
      uint32_t t[N];
      int32_t c = 0;
#pragma unroll
      for (int i = 0; i < N - 1; i++) {
        int32_t d = (int32_t)(a.l[i] - kp<K>(i)) + c;
        t[i] = (uint32_t)d & MASK;
        c = d >> RB;
      }
      int32_t top = (int32_t)(a.l[N - 1] - kp<K>(N - 1)) + c;
      const bool ge = top >= 0;
#pragma unroll
      for (int i = 0; i < N - 1; i++)
        a.l[i] = ge ? t[i] : a.l[i];
      a.l[N - 1] = ge ? (uint32_t)top : a.l[N - 1];
#ifdef BIGFIELD_BOUNDS
      a.bnd = (a.bnd - (double)K > (double)K) ? a.bnd - (double)K : (a.bnd < (double)K ? a.bnd : (double)K);
#endif
    }
    // value < 16p -> < 4p (two conditional subtractions); used by ec.hpp in Fq2Ops' TIGHT mode only
    static HD fe below4(const fe& a)
    {
      BF_ASSERT(a.bnd <= 16.0, "below4 input bound");
      fe r = a;
      cond_sub<8>(r);
      cond_sub<4>(r);
      return r;
    }
    static HD fe reduce(const fe& a)
    { // value < 32p  ->  [0,p)
      BF_ASSERT(a.bnd <= 32.0, "reduce input bound");
      fe r = a;
      cond_sub<16>(r);
      cond_sub<8>(r);
      cond_sub<4>(r);
      cond_sub<2>(r);
      cond_sub<1>(r);
      BF_SET_BOUND(r, 1);
      return r;
    }
    static HD bool is_zero_limbs(const fe& a)
    {
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < N; i++)
        o |= a.l[i];
      return o == 0;
    }
    // exact test a == 0 (mod p) for any in-bound a
    static HD bool is_zero(const fe& a) { return is_zero_limbs(reduce(a)); }
    // Cheap NECESSARY condition for "a == 0 mod p" when a is a Montgomery-product output (< 4p):
    // then a is one of {0,p,2p,3p}, so its lowest limb is one of four constants. False positives
    // (~2^-27) must be followed by is_zero().
    static HD bool maybe_zero_mulout(const fe& a)
    {
      const uint32_t l0 = a.l[0];
      return (l0 == 0) | (l0 == PR::P[0]) | (l0 == ((2 * PR::P[0]) & MASK)) | (l0 == ((3 * PR::P[0]) & MASK));
    }
    static HD bool eq(const fe& a, const fe& b)
    {
      fe ra = reduce(a), rb = reduce(b);
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < N; i++)
        o |= ra.l[i] ^ rb.l[i];
      return o == 0;
    }

    // a^(p-2) (Montgomery in, Montgomery out); a = 0 gives 0. Cold paths only (affine conversion).
    static HD fe inv(const fe& a)
    {
      fe r = one(), base = a;
      uint32_t borrow = 2; // exponent p - 2, word by word (BLS12-377's p ends in ...00000001: the borrow travels)
      for (int wi = 0; wi < N32; wi++) {
        const uint32_t w = PR::P32[wi];
        const uint32_t e = w - borrow;
        borrow = w < borrow ? 1u : 0u;
        for (int b = 0; b < 32; b++) {
          if ((e >> b) & 1) r = mul(r, base);
          base = sqr(base);
        }
      }
      return r;
    }

    // ---- packed 32-bit <-> 29-bit limbs ----
    static HD fe unpack(const uint32_t* w)
    { // w[N32] little-endian words -> normalised limbs (no Montgomery conversion)
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++) {
        const int bit = RB * i;
        const int word = bit / 32, sh = bit % 32;
        uint64_t lo = word < N32 ? w[word] : 0;
        uint64_t hi = (word + 1) < N32 ? w[word + 1] : 0;
        uint32_t v = (uint32_t)(((hi << 32) | lo) >> sh);
        r.l[i] = (i == N - 1) ? v : (v & MASK);
      }
      BF_SET_BOUND(r, 1.2); // caller-supplied canonical value (< 2^NBITS may slightly exceed p)
      return r;
    }
    static HD void pack(uint32_t* w, const fe& a)
    { // a must be canonical ([0,p)), limbs normalised
#pragma unroll
      for (int j = 0; j < N32; j++) {
        const int bit = 32 * j;
        const int i = bit / RB, sh = bit % RB; // word j starts inside limb i at bit sh
        uint64_t v = (uint64_t)a.l[i] >> sh;
        int have = RB - sh;
        if (i + 1 < N) v |= (uint64_t)a.l[i + 1] << have;
        have += RB;
        if (have < 32 && i + 2 < N) v |= (uint64_t)a.l[i + 2] << have;
        w[j] = (uint32_t)v;
      }
    }

    // canonical words -> Montgomery element ; Montgomery element -> canonical words
    static HD fe from_canonical(const uint32_t* w)
    {
      fe r2;
#pragma unroll
      for (int i = 0; i < N; i++)
        r2.l[i] = PR::R2[i];
      BF_SET_BOUND(r2, 1);
      return mul(unpack(w), r2);
    }
    static HD fe from_refmont(const uint32_t* w)
    { // reference Montgomery form (x*2^(32*N32)) -> our Montgomery form
      fe c;
#pragma unroll
      for (int i = 0; i < N; i++)
        c.l[i] = PR::REFMONT_TO_MONT[i];
      BF_SET_BOUND(c, 1);
      return mul(unpack(w), c);
    }
    static HD void to_canonical(uint32_t* w, const fe& a)
    {
      fe o = zero();
      o.l[0] = 1;
      BF_SET_BOUND(o, 1);
      pack(w, reduce(mul(a, o)));
    }
    static HD void to_refmont(uint32_t* w, const fe& a)
    {
      fe c;
#pragma unroll
      for (int i = 0; i < N; i++)
        c.l[i] = PR::MONT_TO_REFMONT[i];
      BF_SET_BOUND(c, 1);
      pack(w, reduce(mul(a, c)));
    }
  };

} // namespace icicle_hip
