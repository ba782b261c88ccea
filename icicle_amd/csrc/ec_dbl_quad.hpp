// Complete projective doubling for the curves whose 3 b is a small integer (BN254: 9, BLS12-381: 12, BLS12-377: 3), laid out for the
// four lanes of a DPP quad -- the doubling of the ECNTT butterflies' chain (ecntt.hip mul_words_quad) from round 6 on.
//
// Renes-Costello-Batina 2016, Algorithm 9 (a = 0), the formula of the reference's dbl (icicle/include/icicle/curves/projective.h:73-99)
// and of ec.hpp dbl_body, has nine products. With 3 b Z^2 formed by additions (3 b is 3, 9 or 12) eight are left, in TWO dependent
// levels of exactly four:
//   level 1:  Y^2 | Y Z | Z^2 | X Y
//   level 2:  (3b Z^2)(8 Y^2) | (Y^2 - 9b Z^2)(Y^2 + 3b Z^2) | (Y Z)(8 Y^2) | (Y^2 - 9b Z^2)(X Y)
// so a quad spends two product latencies and two products per lane on a doubling. The Jacobian chain of rounds 3-6 (dbl_jac_quad:
// 2M + 5S in three levels, the third one product wide) spent three, plus the coordinate changes (X : Y : Z) <-> Jacobian around every
// window (two more levels each way): per four-bit window 16 product latencies, now 8. The formula is complete: the identity
// (0 : 1 : 0) and points of order two need no special case, which the Jacobian form had (Z = 0 -> identity on the way back).
//
// Bounds (units of p, bigfield.hpp; machine-checked by tests/host_math_harness.cpp op 7 under -DBIGFIELD_BOUNDS, which runs dbl() --
// the same operand flow on one lane): inputs X, Y, Z <= 8; outputs X <= 2.5, Y <= 3.3, Z <= 1.2, inside the <= 4 the complete
// addition and the butterfly's negation are laid out for.
#pragma once
#include "ec.hpp"

namespace icicle_hip {

  template <class C>
  struct EcDblSmallB {
    static_assert(C::EXT_DEGREE == 1 && (C::B3_SMALL == 3 || C::B3_SMALL == 9 || C::B3_SMALL == 12), "3 b must be 3, 9 or 12");
    using E = EC<C>;
    using F = typename E::F;
    using fe = typename F::fe;
    using Proj = typename E::Proj;

    // 3 b x by additions, brought back below 4 p (x <= 1.4: a product)
    static HD fe mul_b3(const fe& x)
    {
      fe r;
      if constexpr (C::B3_SMALL == 3)
        r = F::add(F::dbl(x), x);
      else if constexpr (C::B3_SMALL == 9)
        r = F::add(F::dbl(F::dbl(F::dbl(x))), x);
      else
        r = F::dbl(F::dbl(F::add(F::dbl(x), x)));
      if constexpr (C::B3_SMALL == 3) {
        F::template cond_sub<4>(r); // (<= 4.2 p -> below 4 p with one conditional subtraction)
        return r;
      } else
        return F::below4(r);
    }
    struct Mid { // the linear middle part, shared by both forms
      fe z8, t2, y3, t0m;
    };
    static HD Mid middle(const fe& t0, const fe& zz)
    {
      Mid m;
      m.z8 = F::dbl(F::dbl(F::dbl(t0)));                      // 8 Y^2
      m.t2 = mul_b3(zz);                                      // 3b Z^2          < 4
      m.y3 = F::add(t0, m.t2);                                // Y^2 + 3b Z^2
      m.t0m = F::template sub<16>(t0, F::add(F::dbl(m.t2), m.t2)); // Y^2 - 9b Z^2 (subtrahend < 12)
      return m;
    }
    // one lane's arithmetic (host bound checker, tests; the values of ec.hpp dbl_body)
    static HD Proj dbl(const Proj& p)
    {
      const fe t0 = F::sqr(p.y), t1 = F::mul(p.y, p.z), zz = F::sqr(p.z), xy = F::mul(p.x, p.y);
      const Mid m = middle(t0, zz);
      Proj r;
      r.y = F::add(F::mul(m.t2, m.z8), F::mul(m.t0m, m.y3));
      r.z = F::mul(t1, m.z8);
      r.x = F::dbl(F::mul(m.t0m, xy));
      return r;
    }
#if defined(__HIPCC__)
    // the same over a quad: every lane holds the same p, lane `role` computes one product of each level
    static __device__ __forceinline__ Proj dbl_quad(const Proj& p, uint32_t role)
    {
      const bool r0 = role == 0, r1 = role == 1, r2 = role == 2, r3 = role == 3;
      // level 1: role 0: Y Y | role 1: Y Z | role 2: Z Z | role 3: X Y
      const fe a = F::mul(E::lane_select(r3, p.x, E::lane_select(r2, p.z, p.y)), E::lane_select(r0 || r3, p.y, p.z));
      const fe t0 = E::template quad_bcast<0>(a), t1 = E::template quad_bcast<1>(a), zz = E::template quad_bcast<2>(a), xy = E::template quad_bcast<3>(a);
      const Mid m = middle(t0, zz);
      // level 2: role 0: t2 z8 | role 1: t0m y3 | role 2: t1 z8 | role 3: t0m xy
      const fe b = F::mul(E::lane_select(r0, m.t2, E::lane_select(r2, t1, m.t0m)), E::lane_select(r0 || r2, m.z8, E::lane_select(r1, m.y3, xy)));
      Proj r;
      r.y = F::add(E::template quad_bcast<0>(b), E::template quad_bcast<1>(b));
      r.z = E::template quad_bcast<2>(b);
      r.x = F::dbl(E::template quad_bcast<3>(b));
      return r;
    }
#endif
  };

} // namespace icicle_hip
