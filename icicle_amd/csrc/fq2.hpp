// Quadratic extension Fq2 = Fq[u]/(u^2 + NONRES) over bigfield.hpp, for the G2 groups. BN254 and
// BLS12-381 both use the non-residue -1 (icicle/include/icicle/fields/snark_fields/bn254_base.h,
// bls12_381_base.h `nonresidue = 1, nonresidue_is_negative = true`; element layout {c0 = real,
// c1 = imaginary}, icicle/include/icicle/fields/complex_extension.h).
//
// Fq2Ops<PR> offers the same static interface as FieldOps<PR>, so ec.hpp and the MSM kernels are
// instantiated over it unchanged. Lazy bounds are tracked per component (units of p).
//   mul  : c0 = a0*b0 + (16p - a1)*b1 , c1 = a0*b1 + a1*b0  -- two mul_add, i.e. 4 limb products
//          under 2 interleaved reductions (6 N^2 mads; 3-product Karatsuba with separate reductions
//          costs the same and needs more additions)
//   sqr  : c0 = (a0 + a1)*(a0 - a1) , c1 = 2*a0*a1           -- 2 products
// The 16p offset of the negation costs 16*B(b1)/(R/p) in the product's bound. BLS12-381 has
// R/p = 2^25 and does not notice; BN254 has R/p = 128, where products would come out at up to ~3p
// instead of the <= 2p the formulas in ec.hpp were laid out for. TIGHT mode (R/p < 1024) therefore
// ends every product with one conditional subtraction of 2p per component (~6 % of a product), and
// squares with c0 = a0*a0 + (16p - a1)*a1 so that the raw value stays below 4p. With that, Fq2
// products have the same bound class as Fq products and ec.hpp needs no G2-specific constants.
//
// BLS12-377 extends with u^2 = -5 (bls12_377_base.h:1560-1564 `nonresidue = 5`): PR::NONRES = 5 takes the real part
// as a0*b0 - 5*(a1*b1) from two separately reduced products (the factor cannot ride inside an interleaved reduction
// without leaving the operand bounds), brought back below p by four conditional subtractions; the imaginary part is
// the same mul_add as above. Its base field has R/p = 2^29, so every product comes out below (1 + 2^-20) p.
#pragma once
#include "bigfield.hpp"

namespace icicle_hip {

  template <class PR>
  struct Fe2 {
    Fe<PR> c0, c1;
  };

  template <class PR>
  struct Fq2Ops {
    using B = FieldOps<PR>;
    using bfe = typename B::fe;
    using fe = Fe2<PR>;
    static constexpr int N = B::N;           // limbs per component
    static constexpr int N32 = 2 * B::N32;   // packed words per element
    static constexpr int BN32 = B::N32;
    static constexpr bool TIGHT = B::r_over_p() < 1024.0;
    static constexpr int NR = PR::NONRES; // u^2 = -NR
    static_assert(NR == 1 || (NR == 5 && !TIGHT), "Fq2: non-residues other than 1 need the slack of a wide base field");

    // v0 - NR*v1 for two product outputs (NR = 5), brought below p: the formulas of ec.hpp give the products of a
    // field with slack (not TIGHT) a bound of ~1
    static HD bfe sub_nr(const bfe& v0, const bfe& v1)
    {
      const bfe v4 = B::dbl(B::dbl(v1));
      bfe r = B::template sub<8>(v0, B::add(v4, v1));
      B::template cond_sub<8>(r);
      B::template cond_sub<4>(r);
      B::template cond_sub<2>(r);
      B::template cond_sub<1>(r);
      return r;
    }

    static HD void tighten(bfe& x)
    {
      if constexpr (TIGHT) {
        BF_ASSERT(x.bnd <= 4.0, "tighten: raw product not below 4p");
        B::template cond_sub<2>(x);
      }
    }

    static HD fe zero()
    {
      fe r;
      r.c0 = B::zero();
      r.c1 = B::zero();
      return r;
    }
    static HD fe one()
    {
      fe r;
      r.c0 = B::one();
      r.c1 = B::zero();
      return r;
    }
    static HD fe r2() // (R^2, 0): see FieldOps::r2
    {
      fe r;
      r.c0 = B::r2();
      r.c1 = B::zero();
      return r;
    }
    static HD bfe base_r2() { return B::r2(); }
    static HD bfe base_plain_one() { return B::plain_one(); }
    // constant given as [2][N] Montgomery limbs
    template <class ARR>
    static HD fe from_const(const ARR& c)
    {
      fe r;
#pragma unroll
      for (int i = 0; i < N; i++) {
        r.c0.l[i] = c[0][i];
        r.c1.l[i] = c[1][i];
      }
      BF_SET_BOUND(r.c0, 1);
      BF_SET_BOUND(r.c1, 1);
      return r;
    }

    static HD fe mul(const fe& a, const fe& b)
    {
      fe r;
      if constexpr (NR == 1)
        r.c0 = B::mul_add(a.c0, b.c0, B::template neg<16>(a.c1), b.c1);
      else
        r.c0 = sub_nr(B::mul(a.c0, b.c0), B::mul(a.c1, b.c1));
      r.c1 = B::mul_add(a.c0, b.c1, a.c1, b.c0);
      tighten(r.c0);
      tighten(r.c1);
      return r;
    }
    static HD void mul_inplace(fe& a, const fe& b) { a = mul(a, b); }
    static HD void mul_add_inplace_c(fe& c, const fe& a, const fe& b, const fe& d) { c = mul_add(a, b, c, d); }
    static HD fe sqr(const fe& a)
    {
      fe r;
      if constexpr (NR != 1)
        r.c0 = sub_nr(B::sqr(a.c0), B::sqr(a.c1));
      else if constexpr (TIGHT)
        r.c0 = B::mul_add(a.c0, a.c0, B::template neg<16>(a.c1), a.c1);
      else
        r.c0 = B::mul(B::add(a.c0, a.c1), B::template sub<16>(a.c0, a.c1));
      r.c1 = B::mul(B::dbl(a.c0), a.c1);
      tighten(r.c0);
      tighten(r.c1);
      return r;
    }
    static HD fe mul_add(const fe& a, const fe& b, const fe& c, const fe& d) { return add(mul(a, b), mul(c, d)); }
    // multiplication by a base-field element
    static HD fe mul_base(const fe& a, const bfe& s)
    {
      fe r;
      r.c0 = B::mul(a.c0, s);
      r.c1 = B::mul(a.c1, s);
      return r;
    }
    static HD fe add(const fe& a, const fe& b)
    {
      fe r;
      r.c0 = B::add(a.c0, b.c0);
      r.c1 = B::add(a.c1, b.c1);
      return r;
    }
    static HD fe dbl(const fe& a) { return add(a, a); }
    template <int K>
    static HD fe sub(const fe& a, const fe& b)
    {
      fe r;
      r.c0 = B::template sub<K>(a.c0, b.c0);
      r.c1 = B::template sub<K>(a.c1, b.c1);
      return r;
    }
    template <int K>
    static HD fe neg(const fe& a)
    {
      return sub<K>(zero(), a);
    }
    static HD fe select(bool cond, const fe& a, const fe& b)
    {
      fe r;
      r.c0 = B::select(cond, a.c0, b.c0);
      r.c1 = B::select(cond, a.c1, b.c1);
      return r;
    }
    static HD fe below4(const fe& a)
    {
      fe r;
      r.c0 = B::below4(a.c0);
      r.c1 = B::below4(a.c1);
      return r;
    }
    static HD fe reduce(const fe& a)
    {
      fe r;
      r.c0 = B::reduce(a.c0);
      r.c1 = B::reduce(a.c1);
      return r;
    }
    static HD bool is_zero(const fe& a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
    static HD bool maybe_zero_mulout(const fe& a) { return B::maybe_zero_mulout(a.c0) & B::maybe_zero_mulout(a.c1); }
    static HD bool eq(const fe& a, const fe& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }

    // (a0 + a1 u)^-1 = (a0 - a1 u) / (a0^2 + NR a1^2)
    static HD fe inv(const fe& a)
    {
      bfe n1 = B::sqr(a.c1);
      if constexpr (NR == 5) n1 = B::add(B::dbl(B::dbl(n1)), n1);
      const bfe nrm = B::reduce(B::add(B::sqr(a.c0), n1));
      const bfe ni = B::inv(nrm);
      fe r;
      r.c0 = B::mul(a.c0, ni);
      r.c1 = B::mul(B::template neg<16>(a.c1), ni);
      return r;
    }

    static HD fe unpack(const uint32_t* w)
    {
      fe r;
      r.c0 = B::unpack(w);
      r.c1 = B::unpack(w + BN32);
      return r;
    }
    static HD void pack(uint32_t* w, const fe& a)
    {
      B::pack(w, a.c0);
      B::pack(w + BN32, a.c1);
    }
    static HD fe from_canonical(const uint32_t* w)
    {
      fe r;
      r.c0 = B::from_canonical(w);
      r.c1 = B::from_canonical(w + BN32);
      return r;
    }
    static HD fe from_refmont(const uint32_t* w)
    {
      fe r;
      r.c0 = B::from_refmont(w);
      r.c1 = B::from_refmont(w + BN32);
      return r;
    }
    static HD void to_canonical(uint32_t* w, const fe& a)
    {
      B::to_canonical(w, a.c0);
      B::to_canonical(w + BN32, a.c1);
    }
    static HD void to_refmont(uint32_t* w, const fe& a)
    {
      B::to_refmont(w, a.c0);
      B::to_refmont(w + BN32, a.c1);
    }
  };

} // namespace icicle_hip
