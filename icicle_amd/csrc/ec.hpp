// Short-Weierstrass (a = 0) group arithmetic over bigfield.hpp, for the MSM hot path.
//
// The reference does every EC operation with the complete homogeneous-projective formulas of
// Renes-Costello-Batina (icicle/include/icicle/curves/projective.h:73-188; mixed add = 11 muls +
// 2 small-constant muls). The bucket-accumulation hot loop here uses extended-Jacobian "XYZZ"
// accumulators instead (x = X/ZZ, y = Y/ZZZ; mixed add = 8M + 2S, no multiplication by b),
// with the three exceptional inputs (empty accumulator, P+P, P+(-P)) peeled off into rare
// divergent branches. Everything after accumulation (bucket reduction, window combine) is <2 % of
// the work and uses the same complete projective addition as the reference, which has no
// exceptional cases at all (identity = (0:1:0), projective.h:26).
//
// Scaling convention of the hot loop (no Montgomery copy of the bases is ever made): the gathered affine point is
// used AS IT LIES IN HBM -- plain canonical integers (x2, y2) -- and the accumulator keeps X, Y in Montgomery form
// (X*R, Y*R) but ZZ, ZZZ doubly scaled (ZZ*R^2, ZZZ*R^2). Then montmul(x2, ZZ*R^2) = (x2*ZZ)*R is U2 in Montgomery
// form at no cost, every other product of madd-2008-s sees two Montgomery operands, and montmul(ZZ*R^2, PP*R) =
// (ZZ*PP)*R^2 keeps ZZ, ZZZ in their form. The only conversions are per BUCKET, not per point: the first point of
// a bucket is multiplied by R^2 (2 products) and to_proj() removes one R from ZZ (1 product).
//
// Bounds (units of p, see bigfield.hpp; machine-checked under -DBIGFIELD_BOUNDS):
//   affine (plain words) x <= 1.2, y <= 2.2         XYZZ  X <= 8, Y <= 4, ZZ,ZZZ <= 2
//   projective           X,Y,Z <= 4
//
// The same code serves G1 (coordinates in Fq, FieldOps) and G2 (coordinates in Fq2, fq2.hpp):
// C::EXT_DEGREE selects the field-ops class, everything below only uses their common interface.
#pragma once
#include "bigfield.hpp"
#include "fq2.hpp"
#include <type_traits>

#if defined(__HIPCC__)
  #define EC_NOINLINE __host__ __device__ __noinline__
#else
  #define EC_NOINLINE __attribute__((noinline))
#endif

namespace icicle_hip {

  template <class C>
  struct EC {
    using F = std::conditional_t<C::EXT_DEGREE == 2, Fq2Ops<typename C::fq>, FieldOps<typename C::fq>>;
    using fe = typename F::fe;
    static constexpr int N32 = F::N32; // packed words per coordinate

    struct Aff { // Montgomery coordinates; never the identity
      fe x, y;
    };
    struct XYZZ {
      fe x, y, zz, zzz;
    };
    struct Proj {
      fe x, y, z;
    };

    static HD fe b3() { return F::from_const(C::B3); }
    static HD Aff generator()
    {
      Aff g;
      g.x = F::from_const(C::GX);
      g.y = F::from_const(C::GY);
      return g;
    }

    // ---- affine points in memory -------------------------------------------------------------
    // Reference layout Affine<F>{x,y}: 2*N32 words (curves/affine.h:9-34); identity is (0,0).
    static HD bool words_are_zero(const uint32_t* w)
    {
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < 2 * N32; i++)
        o |= w[i];
      return o == 0;
    }
    // affine words as they lie in HBM -> limbs, NO Montgomery conversion (plain integers if the words are canonical)
    static HD Aff load_plain(const uint32_t* w)
    {
      Aff a;
      a.x = F::unpack(w);
      a.y = F::unpack(w + N32);
      return a;
    }
    static HD Aff neg(const Aff& a)
    {
      Aff r;
      r.x = a.x;
      r.y = F::template neg<2>(a.y);
      return r;
    }
    static HD Aff cneg(const Aff& a, bool negate)
    {
      Aff r;
      r.x = a.x;
      r.y = F::select(negate, F::template neg<2>(a.y), a.y);
      return r;
    }

    // ---- XYZZ --------------------------------------------------------------------------------
    // 2*(x,y) for a PLAIN affine point, result in the accumulator's scaling (dbl-2008-s-1 with ZZ1 = ZZZ1 = 1, a = 0).
    // Rare path (a bucket holding the same point twice): 4 extra products for the scalings.
    static HD XYZZ dbl_affine(const Aff& plain)
    {
      XYZZ r;
      Aff p;
      p.x = F::mul_base(plain.x, F::base_r2());
      p.y = F::mul_base(plain.y, F::base_r2());
      fe U = F::dbl(p.y);                // <= 4.4
      fe V = F::sqr(U);                  // ~1.2
      fe W = F::mul(U, V);               // ~1.1
      fe S = F::mul(p.x, V);             // ~1.1
      fe X2 = F::sqr(p.x);               // ~1.1
      fe M = F::add(F::dbl(X2), X2);     // 3x^2 <= 3.3
      fe S2 = F::dbl(S);                 // <= 2.2
      r.x = F::template sub<4>(F::sqr(M), S2);             // <= 5.1
      if constexpr (F::TIGHT) r.x = F::below4(r.x);        // Fq2 over BN254: stored X stays < 4p
      fe t = F::template sub<8>(S, r.x);                   // <= 9.1
      r.y = F::template sub<2>(F::mul(M, t), F::mul(W, p.y)); // <= 3.3
      r.zz = F::mul_base(V, F::base_r2());
      r.zzz = F::mul_base(W, F::base_r2());
      return r;
    }

    // acc += b (b = PLAIN affine limbs, not the identity). `empty` is the accumulator's "is identity" flag.
    // madd-2008-s: 8M + 2S. Scaling: see the file header.
    static HD void madd(XYZZ& acc, bool& empty, const Aff& b)
    {
      if (__builtin_expect(empty, 0)) {
        acc.x = F::mul_base(b.x, F::base_r2());
        acc.y = F::mul_base(b.y, F::base_r2());
        acc.zz = F::r2();
        acc.zzz = F::r2();
        empty = false;
        return;
      }
      fe U2 = F::mul(b.x, acc.zz);
      fe S2 = F::mul(b.y, acc.zzz);
      // TIGHT (Fq2 over BN254, products < 2p rather than ~1.2p): X is kept < 4p, so 4p suffices here
      fe P = F::template sub<(F::TIGHT ? 4 : 8)>(U2, acc.x); // <= 9.1
      fe R = F::template sub<4>(S2, acc.y);                  // <= 5.1
      fe PP = F::sqr(P);                    // <= 1.7
      if (__builtin_expect(F::maybe_zero_mulout(PP), 0)) {
        if (F::is_zero(P)) { // same x: either b == acc (double) or b == -acc (cancel)
          if (F::is_zero(R)) {
            acc = dbl_affine(b);
          } else {
            empty = true;
          }
          return;
        }
      }
      fe PPP = F::mul(P, PP);               // ~1.2
      fe Q = F::mul(acc.x, PP);             // ~1.2
      fe t = F::add(PPP, F::dbl(Q));        // <= 3.6
      fe X3 = F::template sub<(F::TIGHT ? 8 : 4)>(F::sqr(R), t); // <= 5.3
      if constexpr (F::TIGHT) X3 = F::below4(X3);
      fe d = F::template sub<(F::TIGHT ? 4 : 8)>(Q, X3);         // <= 9.2
      // Y3 = R*d - Y1*PPP = R*d + (4p - Y1)*PPP with one shared reduction (lazy, mul_add), produced in Y's own
      // registers; ZZ, ZZZ likewise (no copies of the accumulator at the loop's back edge)
#ifndef EC_NO_INPLACE
      acc.y = F::template neg<4>(acc.y);
      F::mul_add_inplace_c(acc.y, R, d, PPP); // <= 1.5
      F::mul_inplace(acc.zz, PP);
      F::mul_inplace(acc.zzz, PPP);
      acc.x = X3;
#else // A/B switch for tools/ab_lib.sh
      fe Y3 = F::mul_add(R, d, F::template neg<4>(acc.y), PPP);
      acc.zz = F::mul(acc.zz, PP);
      acc.zzz = F::mul(acc.zzz, PPP);
      acc.x = X3;
      acc.y = Y3;
#endif
    }

    // ---- complete projective arithmetic (identity = (0:1:0)) -----------------------------------
    static HD Proj proj_identity()
    {
      Proj r;
      r.x = F::zero();
      r.y = F::one();
      r.z = F::zero();
      return r;
    }
    // (X*R, Y*R, ZZ*R^2, ZZZ*R^2) -> (X*ZZZ : Y*ZZ : ZZ*ZZZ), every coordinate scaled by the same R^2, i.e. the
    // Montgomery form of the projective point (X*ZZZ*R : Y*ZZ*R : ZZ*ZZZ*R) = the same point
    static HD Proj to_proj(const XYZZ& a, bool empty)
    {
      if (empty) return proj_identity();
      Proj r;
      const fe zz1 = F::mul_base(a.zz, F::base_plain_one()); // ZZ*R
      r.x = F::mul(a.x, a.zzz);
      r.y = F::mul(a.y, a.zz);
      r.z = F::mul(zz1, a.zzz);
      return r;
    }
    static HD Proj to_proj(const Aff& a)
    {
      Proj r;
      r.x = a.x;
      r.y = a.y;
      r.z = F::one();
      return r;
    }
    // Renes-Costello-Batina 2016, Algorithm 7 (a = 0), the formula the reference uses
    // (projective.h:101-143): 12M + 2 mul-by-3b, valid for ALL inputs.
    // Over Fq2 the body is ~5000 (BN254) to ~12000 (BLS12-381) instructions; the cold kernels that use
    // it call it as a function there (one copy per kernel) instead of inlining it at every site.
    static HD Proj add(const Proj& p, const Proj& q)
    {
      if constexpr (C::EXT_DEGREE == 2)
        return add_call(p, q);
      else
        return add_body(p, q);
    }
    static EC_NOINLINE Proj add_call(const Proj& p, const Proj& q) { return add_body(p, q); }
    static HD Proj add_body(const Proj& p, const Proj& q)
    {
      fe t0 = F::mul(p.x, q.x);
      fe t1 = F::mul(p.y, q.y);
      fe t2 = F::mul(p.z, q.z);
      fe t3 = F::mul(F::add(p.x, p.y), F::add(q.x, q.y));
      t3 = F::template sub<4>(t3, F::add(t0, t1)); // X1Y2 + X2Y1
      fe t4 = F::mul(F::add(p.y, p.z), F::add(q.y, q.z));
      t4 = F::template sub<4>(t4, F::add(t1, t2)); // Y1Z2 + Y2Z1
      fe t5 = F::mul(F::add(p.x, p.z), F::add(q.x, q.z));
      t5 = F::template sub<4>(t5, F::add(t0, t2)); // X1Z2 + X2Z1
      fe t0_3 = F::add(F::dbl(t0), t0);            // 3 X1X2
      fe bt2 = F::mul(b3(), t2);
      fe z3 = F::add(t1, bt2);
      fe t1m = F::template sub<2>(t1, bt2);
      fe y3 = F::mul(b3(), t5);
      Proj r;
      r.x = F::template sub<2>(F::mul(t3, t1m), F::mul(t4, y3));
      r.y = F::add(F::mul(t1m, z3), F::mul(y3, t0_3));
      r.z = F::add(F::mul(z3, t4), F::mul(t0_3, t3));
      return r;
    }
    // Renes-Costello-Batina 2016, Algorithm 9 (a = 0), the reference's dbl (projective.h:73-99):
    // 6M + 2S + 1 mul-by-3b instead of the 14 of add(p, p).
    static HD Proj dbl(const Proj& p)
    {
      if constexpr (C::EXT_DEGREE == 2)
        return dbl_call(p);
      else
        return dbl_body(p);
    }
    static EC_NOINLINE Proj dbl_call(const Proj& p) { return dbl_body(p); }
    static HD Proj dbl_body(const Proj& p)
    {
      fe t0 = F::sqr(p.y);                 // Y^2
      fe z8 = F::dbl(F::dbl(F::dbl(t0)));  // 8Y^2            <= 8*1.2
      fe t1 = F::mul(p.y, p.z);            // YZ
      fe t2 = F::mul(b3(), F::sqr(p.z));   // 3b Z^2
      Proj r;
      fe x3 = F::mul(t2, z8);
      fe y3 = F::add(t0, t2);
      r.z = F::mul(t1, z8);
      fe t2_3 = F::add(F::dbl(t2), t2);    // 9b Z^2
      fe t0m = F::template sub<4>(t0, t2_3); // Y^2 - 9b Z^2   <= 5.2
      r.y = F::add(x3, F::mul(t0m, y3));
      r.x = F::dbl(F::mul(t0m, F::mul(p.x, p.y)));
      return r;
    }

    // ---- Jacobian doubling chain (x = X/Z^2, y = Y/Z^3) -------------------------------------------
    // The window combine multiplies each window sum by 2^(c*w): up to ~250 SEQUENTIAL doublings on one
    // lane, the latency floor of a small MSM. dbl-2009-l (a = 0) costs 2M + 5S against 6M + 2S + 1 for
    // the complete projective doubling. Not complete: Z = 0 stays Z = 0 and from_jac() maps it (and
    // only it) back to the identity, which also covers a point of order two (Y = 0 -> Z3 = 0).
    // Coordinates are kept below 4p (below4) so the same constants hold for Fq and for Fq2 in TIGHT mode.
    struct Jac {
      fe x, y, z;
    };
    static HD Jac to_jac(const Proj& p) // (X : Y : Z) -> (X Z : Y Z^2 : Z)
    {
      Jac r;
      fe zz = F::sqr(p.z);
      r.x = F::mul(p.x, p.z);
      r.y = F::mul(p.y, zz);
      r.z = p.z;
      return r;
    }
    static HD Proj from_jac(const Jac& j) // (X : Y : Z) -> (X Z : Y : Z^3)
    {
      if (F::is_zero(j.z)) return proj_identity();
      Proj r;
      fe zz = F::sqr(j.z);
      r.x = F::mul(j.x, j.z);
      r.y = j.y;
      r.z = F::mul(zz, j.z);
      return r;
    }
    static HD Jac dbl_jac(const Jac& p)
    {
      fe A = F::sqr(p.x);
      fe B = F::sqr(p.y);
      fe CC = F::sqr(B);
      fe t = F::sqr(F::add(p.x, B));
      fe D = F::below4(F::dbl(F::template sub<4>(t, F::add(A, CC)))); // 2((X + B)^2 - A - C)
      fe E = F::add(F::dbl(A), A);                                     // 3 X^2
      fe Fv = F::sqr(E);
      Jac r;
      r.x = F::below4(F::template sub<8>(Fv, F::dbl(D)));             // F - 2D
      fe m = F::mul(E, F::template sub<4>(D, r.x));
      fe c8 = F::dbl(F::below4(F::dbl(F::dbl(CC))));                    // 8 Y^4
      r.y = F::below4(F::template sub<8>(m, c8));
      r.z = F::dbl(F::mul(p.y, p.z));
      return r;
    }

    // The same doubling for THROUGHPUT chains over a prime field (msm_precompute_bases: one chain per lane, millions of lanes):
    // a third of dbl_jac's instructions are range management, not products -- four below4() = eight conditional subtractions of
    // ~40 instructions each beside 954 multiply-adds. Over Fq (R / p >= 128, products accept operands up to 64 p) only D needs
    // to come back below 4 p; the coordinates then live in X < 9.2 p, Y < 17.8 p, Z < 2.8 p, a fixed point of the step (walked
    // through by the bound tracker in tests/host_math_harness.cpp, op 6):
    //   A = X^2 < 1.7, B = Y^2 < 3.5, C = B^2 < 1.1, t = (X + B)^2 < 2.3, D = 2 (t - A - C + 4p) < 12.6 -> below4,
    //   E = 3A < 5, F = E^2 < 1.2, X3 = F - 2D + 8p < 9.2, Y3 = E (D - X3 + 16p) - 8C + 16p < 17.8, Z3 = 2 Y Z < 2.8.
    // Not for Fq2 (its products want operands below 16 p, BN254's TIGHT mode below 4 p: dbl_jac); the caller reduces the results
    // (F::reduce takes < 32 p).
    static HD Jac dbl_jac_lazy(const Jac& p)
    {
      if constexpr (C::EXT_DEGREE == 2) { // (Fq2 products take their operands below 16 p, BN254's below 4 p)
        return dbl_jac(p);
      } else {
        fe A = F::sqr(p.x);
        fe B = F::sqr(p.y);
        fe CC = F::sqr(B);
        fe t = F::sqr(F::add(p.x, B));
        fe D = F::below4(F::dbl(F::template sub<4>(t, F::add(A, CC))));
        fe E = F::add(F::dbl(A), A);
        fe Fv = F::sqr(E);
        Jac r;
        r.x = F::template sub<8>(Fv, F::dbl(D));
        fe m = F::mul(E, F::template sub<16>(D, r.x));
        r.y = F::template sub<16>(m, F::dbl(F::dbl(F::dbl(CC))));
        r.z = F::dbl(F::mul(p.y, p.z));
        return r;
      }
    }

#if defined(__HIPCC__)
    // ---- doubling spread over the four lanes of a DPP quad (k_final's Horner chains) -----------------------------
    // A chain of ~250 dependent doublings on ONE wave is pure latency (7 dependent products per step, ~4.4 us each on
    // gfx950: the floor of every small MSM). The 2M + 5S of dbl_jac have depth 3: {X^2, Y^2, YZ} -> {(3X^2)^2, Y^4,
    // (X + Y^2)^2} -> E (D - X3). All four lanes of a quad hold the same point; lane `role` computes one product of
    // each level, the results are broadcast with quad_perm DPP moves, and the cheap linear steps run redundantly.
    // Same operands, same bounds and the same values as dbl_jac, three product latencies per step instead of seven.
    template <int SRC>
    static __device__ __forceinline__ fe quad_bcast(const fe& v)
    {
      constexpr int NW = sizeof(fe) / 4;
      uint32_t w[NW];
      __builtin_memcpy(w, &v, sizeof(fe));
#pragma unroll
      for (int i = 0; i < NW; i++)
        w[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)w[i], (int)w[i], SRC * 0x55, 0xF, 0xF, false); // quad_perm [SRC x 4]
      fe r = v; // keeps the (host-checker-only) bound annotation
      __builtin_memcpy(&r, w, sizeof(fe));
      return r;
    }
    static __device__ __forceinline__ fe lane_select(bool c, const fe& a, const fe& b)
    {
      constexpr int NW = sizeof(fe) / 4;
      uint32_t x[NW], y[NW];
      __builtin_memcpy(x, &a, sizeof(fe));
      __builtin_memcpy(y, &b, sizeof(fe));
#pragma unroll
      for (int i = 0; i < NW; i++)
        x[i] = c ? x[i] : y[i];
      fe r = a;
      __builtin_memcpy(&r, x, sizeof(fe));
      return r;
    }
    static __device__ __forceinline__ Jac dbl_jac_quad(const Jac& p, uint32_t role)
    {
      // level 1: role 0: X*X, role 1: Y*Y, roles 2, 3: Y*Z
      const fe l1 = F::mul(lane_select(role == 0, p.x, p.y), lane_select(role == 0, p.x, lane_select(role == 1, p.y, p.z)));
      const fe A = quad_bcast<0>(l1), B = quad_bcast<1>(l1), YZ = quad_bcast<2>(l1);
      // level 2: role 0: E^2 (E = 3A), role 1: B^2, roles 2, 3: (X + B)^2
      const fe E = F::add(F::dbl(A), A);
      const fe l2 = F::sqr(lane_select(role == 0, E, lane_select(role == 1, B, F::add(p.x, B))));
      const fe Fv = quad_bcast<0>(l2), CC = quad_bcast<1>(l2), t = quad_bcast<2>(l2);
      const fe D = F::below4(F::dbl(F::template sub<4>(t, F::add(A, CC))));
      Jac r;
      r.x = F::below4(F::template sub<8>(Fv, F::dbl(D)));
      // level 3 (every lane): E (D - X3)
      const fe m = F::mul(E, F::template sub<4>(D, r.x));
      const fe c8 = F::dbl(F::below4(F::dbl(F::dbl(CC))));
      r.y = F::below4(F::template sub<8>(m, c8));
      r.z = F::dbl(YZ);
      return r;
    }
    // ---- complete addition spread over the four lanes of a DPP quad (ECNTT butterflies) ----------------------------------
    // add_body has 14 products in three dependent levels (6, 2, 6). All four lanes of a quad hold the same p and q; a lane
    // computes one product per round, results are broadcast with quad_perm moves: five product latencies per addition
    // instead of fourteen. Same operands, bounds and values as add_body (Renes-Costello-Batina Algorithm 7, a = 0).
    static __device__ __forceinline__ Proj add_quad(const Proj& p, const Proj& q, uint32_t role)
    {
      const bool r0 = role == 0, r1 = role == 1, r2 = role == 2;
      const bool even = (role & 1u) == 0;
      const fe sxy1 = F::add(p.x, p.y), sxy2 = F::add(q.x, q.y);
      const fe syz1 = F::add(p.y, p.z), syz2 = F::add(q.y, q.z);
      const fe sxz1 = F::add(p.x, p.z), sxz2 = F::add(q.x, q.z);
      // round A: X1 X2 | Y1 Y2 | Z1 Z2 | (X1 + Y1)(X2 + Y2)
      const fe a = F::mul(lane_select(r0, p.x, lane_select(r1, p.y, lane_select(r2, p.z, sxy1))), lane_select(r0, q.x, lane_select(r1, q.y, lane_select(r2, q.z, sxy2))));
      // round B: (Y1 + Z1)(Y2 + Z2) on the even lanes, (X1 + Z1)(X2 + Z2) on the odd ones
      const fe b = F::mul(lane_select(even, syz1, sxz1), lane_select(even, syz2, sxz2));
      const fe t0 = quad_bcast<0>(a), t1 = quad_bcast<1>(a), t2 = quad_bcast<2>(a), m3 = quad_bcast<3>(a);
      const fe m4 = quad_bcast<0>(b), m5 = quad_bcast<1>(b);
      const fe t3 = F::template sub<4>(m3, F::add(t0, t1)); // X1Y2 + X2Y1
      const fe t4 = F::template sub<4>(m4, F::add(t1, t2)); // Y1Z2 + Y2Z1
      const fe t5 = F::template sub<4>(m5, F::add(t0, t2)); // X1Z2 + X2Z1
      const fe t0_3 = F::add(F::dbl(t0), t0);               // 3 X1X2
      // round C: 3b Z1Z2 on the even lanes, 3b (X1Z2 + X2Z1) on the odd ones
      const fe c = F::mul(b3(), lane_select(even, t2, t5));
      const fe bt2 = quad_bcast<0>(c), y3 = quad_bcast<1>(c);
      const fe z3 = F::add(t1, bt2);
      const fe t1m = F::template sub<2>(t1, bt2);
      // round D: t3 t1m | t4 y3 | t1m z3 | y3 t0_3
      const fe d = F::mul(lane_select(r0, t3, lane_select(r1, t4, lane_select(r2, t1m, y3))), lane_select(r0, t1m, lane_select(r1, y3, lane_select(r2, z3, t0_3))));
      // round E: z3 t4 on the even lanes, t0_3 t3 on the odd ones
      const fe e = F::mul(lane_select(even, z3, t0_3), lane_select(even, t4, t3));
      Proj r;
      r.x = F::template sub<2>(quad_bcast<0>(d), quad_bcast<1>(d));
      r.y = F::add(quad_bcast<2>(d), quad_bcast<3>(d));
      r.z = F::add(quad_bcast<0>(e), quad_bcast<1>(e));
      return r;
    }
#endif

    // k * p for a small unsigned k (bucket reduction segment offsets), MSB-first double-and-add
    static HD Proj mul_small(const Proj& p, uint32_t k)
    {
      Proj r = proj_identity();
      bool started = false;
      for (int bit = 31; bit >= 0; bit--) {
        if (started) r = dbl(r);
        if ((k >> bit) & 1) {
          r = started ? add(r, p) : p;
          started = true;
        }
      }
      return r;
    }

    // Projective{x,y,z} -> reference layout (3*N32 canonical words, non-Montgomery)
    static HD void store_proj_canonical(uint32_t* w, const Proj& p)
    {
      F::to_canonical(w, p.x);
      F::to_canonical(w + N32, p.y);
      F::to_canonical(w + 2 * N32, p.z);
    }
    static HD void store_proj_refmont(uint32_t* w, const Proj& p)
    {
      F::to_refmont(w, p.x);
      F::to_refmont(w + N32, p.y);
      F::to_refmont(w + 2 * N32, p.z);
    }
  };

} // namespace icicle_hip
