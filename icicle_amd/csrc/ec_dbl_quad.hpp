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

  // true for the G1 curves whose 3 b is 3, 9 or 12 (the G2 parameter sets have no B3_SMALL at all)
  template <class C, class = void>
  struct has_small_b3 : std::false_type {
  };
  template <class C>
  struct has_small_b3<C, std::void_t<decltype(C::B3_SMALL)>> : std::bool_constant<C::EXT_DEGREE == 1 && C::B3_SMALL != 0> {
  };

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

#if defined(__HIPCC__)
  // ---- complete addition over a quad in FOUR product rounds -------------------------------------------------------------------
  // ec.hpp add_quad follows add_body's three levels (6, 2, 6 products) literally: five rounds, two of them half empty. The fourteen
  // products pack into 4 + 4 + 4 + 2 when a product starts as soon as its operands exist:
  //   A: X1X2 | Y1Y2 | Z1Z2 | (X1+Y1)(X2+Y2)                 -> t0, t1, t2, t3, 3 t0
  //   B: (Y1+Z1)(Y2+Z2) | (X1+Z1)(X2+Z2) | 3b t2 | (3 t0) t3   -> t4, t5, z3 = t1 + 3b t2, t1m = t1 - 3b t2
  //   C: 3b t5 | t3 t1m | t1m z3 | z3 t4                       -> y3
  //   D: t4 y3 (even lanes) | y3 (3 t0) (odd lanes)
  // Same operand pairs, bounds and values as add_body (Renes-Costello-Batina Algorithm 7, a = 0; the reference's operator+,
  // projective.h:101-143); four products per lane and addition instead of five.
  template <class C>
  struct EcQuadAdd {
    using E = EC<C>;
    using F = typename E::F;
    using fe = typename F::fe;
    using Proj = typename E::Proj;
    static __device__ __forceinline__ fe sel4(uint32_t role, const fe& a0, const fe& a1, const fe& a2, const fe& a3)
    {
      return E::lane_select(role == 0, a0, E::lane_select(role == 1, a1, E::lane_select(role == 2, a2, a3)));
    }
    static __device__ __forceinline__ Proj add(const Proj& p, const Proj& q, uint32_t role)
    {
      const bool even = (role & 1u) == 0;
      const fe b3 = E::b3();
      const fe a = F::mul(sel4(role, p.x, p.y, p.z, F::add(p.x, p.y)), sel4(role, q.x, q.y, q.z, F::add(q.x, q.y)));
      const fe t0 = E::template quad_bcast<0>(a), t1 = E::template quad_bcast<1>(a), t2 = E::template quad_bcast<2>(a);
      const fe t3 = F::template sub<4>(E::template quad_bcast<3>(a), F::add(t0, t1)); // X1Y2 + X2Y1
      const fe t0_3 = F::add(F::dbl(t0), t0);                                          // 3 X1X2
      const fe b = F::mul(sel4(role, F::add(p.y, p.z), F::add(p.x, p.z), b3, t0_3), sel4(role, F::add(q.y, q.z), F::add(q.x, q.z), t2, t3));
      const fe t4 = F::template sub<4>(E::template quad_bcast<0>(b), F::add(t1, t2)); // Y1Z2 + Y2Z1
      const fe t5 = F::template sub<4>(E::template quad_bcast<1>(b), F::add(t0, t2)); // X1Z2 + X2Z1
      const fe bt2 = E::template quad_bcast<2>(b), t03t3 = E::template quad_bcast<3>(b);
      const fe z3 = F::add(t1, bt2);
      const fe t1m = F::template sub<2>(t1, bt2);
      const fe c = F::mul(sel4(role, b3, t3, t1m, z3), sel4(role, t5, t1m, z3, t4));
      const fe y3 = E::template quad_bcast<0>(c);
      const fe d = F::mul(y3, E::lane_select(even, t4, t0_3));
      Proj r;
      r.x = F::template sub<2>(E::template quad_bcast<1>(c), E::template quad_bcast<0>(d));
      r.y = F::add(E::template quad_bcast<2>(c), E::template quad_bcast<1>(d));
      r.z = F::add(E::template quad_bcast<3>(c), t03t3);
      return r;
    }
  };
#endif

} // namespace icicle_hip
