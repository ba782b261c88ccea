// ECNTT for gfx950: the NTT whose elements are G1 points (projective_t) and whose twiddles are
// scalars -- X[k] = sum_j w^(jk) * P_j.
//
// Reference: src/ecntt.cpp:5-18 (<curve>_ecntt); the CPU backend runs its generic NTT with
// E = projective_t, S = scalar_t (backend/cpu/src/curve/cpu_ecntt.cpp:13-20), so every NTTConfig
// feature applies: orderings, coset generator, batch / columns_batch, 1/N on the inverse. It uses the
// twiddle domain of the scalar-field NTT (<curve>_ntt_init_domain).
//
// A butterfly here is one 254-bit scalar multiplication (rounds 3-5: 4-bit windows, 252 Jacobian doublings + <= 77 complete
// additions, ~0.8 M instructions; round 6: GLV split, 27 signed 5-bit joint windows, 130 complete projective doublings + <= 69
// additions -- mul_words_quad below) plus two point additions; memory traffic is irrelevant by five orders of
// magnitude, and a stage of an ECNTT of practical size has far fewer butterflies than the chip has lanes: the
// time of a stage is the LATENCY of one scalar multiplication. So the structure is radix-2 DIT with one launch per
// stage, points in HBM in the kernels' internal form (Montgomery limbs, 3 x 9 or 3 x 14 words), and FOUR lanes (a DPP
// quad) per butterfly sharing the doubling chain (two dependent products per doubling, ec_dbl_quad.hpp); the
// reorderings / coset / 1/N factors are folded into the load and store kernels. The projective representative of a result differs from
// the reference's (it depends on the order of additions); the group element is the same -- tests
// compare to_affine() limbs, as for the MSM.
#include "ntt_big_common.hpp"
#include "ec.hpp"
#include "glv.hpp"
#include "ec_dbl_quad.hpp"
#include "ntt_plan.h"
#include <algorithm>

// the quad addition of every kernel here: four product rounds (ec_dbl_quad.hpp EcQuadAdd); -DECNTT_ADD5 = ec.hpp add_quad's five, for A/B builds
#ifdef ECNTT_ADD5
  #define QADD(a, b, role) E::add_quad(a, b, role)
#else
  #define QADD(a, b, role) EcQuadAdd<C>::add(a, b, role)
#endif

namespace icicle_hip {

  template <class C>
  struct EcNtt {
    using E = EC<C>;
    using F = typename E::F;
    using FR = FieldOps<typename C::fr>;
    using Proj = typename E::Proj;

    // k * p, k = 8 canonical words, MSB-first double-and-add over the complete formulas: the cold uses (coset factors on
    // the way in, 1/N and coset factors on the way out: n scalar multiplications per transform against n/2 log n in the
    // butterflies)
    static __device__ Proj mul_words_serial(const Proj& p, const uint32_t* k)
    {
      Proj r = E::proj_identity();
      bool started = false;
      for (int bit = 255; bit >= 0; bit--) {
        if (started) r = E::dbl(r);
        if ((k[bit >> 5] >> (bit & 31)) & 1) {
          r = started ? E::add(r, p) : p;
          started = true;
        }
      }
      return r;
    }
    // The butterflies' k * p: fixed windows, most significant first (the reference's own scalar multiplication is windowed
    // too, include/icicle/curves/projective.h:192-224), a run of doublings and COMPLETE additions of table entries per window, so
    // no input is exceptional. Three generations live here, the older ones behind A/B macros (tools/ab_lib.sh):
    //   rounds 3-5 (-DECNTT_NO_GLV): 63 four-bit windows, 252 Jacobian doublings (2M + 5S, three product levels per quad) + <= 77 additions;
    //   round 6a (-DECNTT_JAC_DBL / -DECNTT_WIN4): the GLV split, 33 joint four-bit windows, 132 doublings + <= 80 additions;
    //   round 6b (default, curves with small 3 b): GLV, 27 joint signed five-bit windows, 130 complete projective doublings in two
    //   product levels each (ec_dbl_quad.hpp) + <= 69 additions.
    // The four lanes of a DPP quad hold the same operands and share every doubling and addition -- a butterfly is a latency chain
    // and an ECNTT stage of practical size has fewer butterflies than
    // the chip has lanes, so spending four lanes on one chain is free. `tab` = 16 entries of LDS owned by the quad (a
    // private array indexed by the digit would live in scratch memory). Every lane of the WAVE must call this together
    // (the table is published with a barrier); lanes with go == false compute on the identity and the result is unused.
    // (round 6a) dbl_jac_quad (ec.hpp) with the range management of dbl_jac_lazy: inside a run of doublings only D is brought back below 4 p
    // (X < 9.2 p, Y < 17.8 p, Z < 2.8 p is a fixed point of the step), two conditional subtractions per step instead of eight. In quad
    // form every lane executes the linear steps redundantly, so they were ~45 % of a doubling's instructions. The caller reduces Y once
    // per window. The same operand flow, one lane's arithmetic, runs under the host bound tracker: tests/host_math_harness.cpp op 7.
    static __device__ __forceinline__ typename E::Jac dbl_jac_quad_lazy(const typename E::Jac& p, uint32_t role)
    {
      using fe = typename F::fe;
      const fe l1 = F::mul(E::lane_select(role == 0, p.x, p.y), E::lane_select(role == 0, p.x, E::lane_select(role == 1, p.y, p.z)));
      const fe A = E::template quad_bcast<0>(l1), B = E::template quad_bcast<1>(l1), YZ = E::template quad_bcast<2>(l1);
      const fe Ev = F::add(F::dbl(A), A);
      const fe l2 = F::sqr(E::lane_select(role == 0, Ev, E::lane_select(role == 1, B, F::add(p.x, B))));
      const fe Fv = E::template quad_bcast<0>(l2), CC = E::template quad_bcast<1>(l2), t = E::template quad_bcast<2>(l2);
      const fe D = F::below4(F::dbl(F::template sub<4>(t, F::add(A, CC))));
      typename E::Jac r;
      r.x = F::template sub<8>(Fv, F::dbl(D));
      const fe m = F::mul(Ev, F::template sub<16>(D, r.x));
      r.y = F::template sub<16>(m, F::dbl(F::dbl(F::dbl(CC))));
      r.z = F::dbl(YZ);
      return r;
    }
    // The two coordinate changes around a window's doublings, (X : Y : Z) -> (X Z, Y Z^2, Z) and back (X Z : Y : Z^3), are three products
    // each; spread over the quad they are two dependent products per lane instead of three (ec.hpp to_jac / from_jac, same values).
    static __device__ __forceinline__ typename E::Jac to_jac_quad(const Proj& p, uint32_t role)
    {
      using fe = typename F::fe;
      const fe l1 = F::mul(E::lane_select(role == 0, p.x, p.z), p.z); // role 0: X Z ; others: Z^2
      typename E::Jac r;
      r.x = E::template quad_bcast<0>(l1);
      r.y = F::mul(p.y, E::template quad_bcast<1>(l1));
      r.z = p.z;
      return r;
    }
    static __device__ __forceinline__ Proj from_jac_quad(const typename E::Jac& j, uint32_t role)
    {
      using fe = typename F::fe;
      if (F::is_zero(j.z)) return E::proj_identity(); // (uniform over the quad: every lane holds the same point)
      const fe l1 = F::mul(E::lane_select(role == 0, j.x, j.z), j.z); // role 0: X Z ; others: Z^2
      Proj r;
      r.x = E::template quad_bcast<0>(l1);
      r.y = j.y;
      r.z = F::mul(E::template quad_bcast<1>(l1), j.z);
      return r;
    }
    static __device__ Proj mul_words_quad(const Proj& p, const uint32_t* k, uint32_t role, Proj* tab)
    {
      // the additions are quad-cooperative as well (five product latencies instead of fourteen): -DECNTT_NOQUADADD = A/B
#ifdef ECNTT_NOQUADADD
      auto ADD = [&](const Proj& a, const Proj& b) { return E::add(a, b); };
#else
      auto ADD = [&](const Proj& a, const Proj& b) { return QADD(a, b, role); };
#endif
#if !defined(ECNTT_NO_GLV) && !defined(ECNTT_JAC_DBL) && !defined(ECNTT_NOQUAD) && !defined(ECNTT_NO_LAZY_DBL) && !defined(ECNTT_WIN4)
      constexpr bool WIN5 = C::B3_SMALL != 0; // signed five-bit windows over the multiples 1..16 (below); -DECNTT_WIN4 = A/B
#else
      constexpr bool WIN5 = false;
#endif
      if constexpr (WIN5) {
        Proj e = p;
        for (int i = 0; i < 16; i++) { // tab[i] = (i + 1) p
          if (role == 0) tab[i] = e;
          if (i < 15) e = (i == 0) ? EcDblSmallB<C>::dbl_quad(p, role) : ADD(e, p);
        }
      } else {
        Proj e = E::proj_identity();
        for (int i = 0; i < 16; i++) {
          if (role == 0) tab[i] = e;
          e = (i == 0) ? p : ((i == 1) ? E::dbl(p) : ADD(e, p)); // 0, p, 2p, 3p, ...
        }
      }
      __syncthreads();
      auto quad_dbl4 = [&](Proj& r) {
#if !defined(ECNTT_JAC_DBL) && !defined(ECNTT_NOQUAD) && !defined(ECNTT_NO_LAZY_DBL) // (-DECNTT_JAC_DBL: the Jacobian chain below, for A/B builds)
        if constexpr (C::B3_SMALL != 0) {
          // complete projective doublings in two product levels each, no coordinate change around the window (ec_dbl_quad.hpp)
          for (int q = 0; q < 4; q++)
            r = EcDblSmallB<C>::dbl_quad(r, role);
          return;
        }
#endif
#if defined(ECNTT_NOQUAD) || defined(ECNTT_NO_LAZY_DBL)
        typename E::Jac j = E::to_jac(r);
#else
        typename E::Jac j = to_jac_quad(r, role);
#endif
#ifdef ECNTT_NOQUAD
        for (int q = 0; q < 4; q++)
          j = E::dbl_jac(j);
#elif defined(ECNTT_NO_LAZY_DBL) // (A/B: the fully reduced quad doubling of rounds 3-5)
        for (int q = 0; q < 4; q++)
          j = E::dbl_jac_quad(j, role);
#else
        for (int q = 0; q < 4; q++)
          j = dbl_jac_quad_lazy(j, role);
        F::template cond_sub<16>(j.y); // back to Y < 4 p, the bound the complete addition (and the butterfly's negation) is laid out for
        j.y = F::below4(j.y);
        r = from_jac_quad(j, role);
        return;
#endif
        r = E::from_jac(j);
      };
      Proj r = E::proj_identity();
      bool started = false;
#ifndef ECNTT_NO_GLV // (-DECNTT_NO_GLV: the plain 63-window chain of rounds 3-5, for A/B builds with tools/ab_lib.sh)
      {
        // Round 6: k P = k1 P + k2 phi(P), phi(x, y) = (beta x, y), |k1|, |k2| < 2^129 (glv.hpp) -- ONE joint chain of 33 windows
        // (128 + 4 doublings) with up to two additions each instead of 63 windows (252 doublings) with one: the multiples of
        // phi(P) are the same table entries with X scaled by beta, a negative half negates the entry's Y. The chain is the latency
        // of a stage and, from 2^16 points up, its throughput: 252 x 3 + 78 x 5 dependent product latencies become 132 x 3 + 81 x 5.
        uint32_t k1[5], k2[5];
        bool n1, n2;
        glv_decompose<C>(k, k1, n1, k2, n2);
        const typename F::fe beta = F::from_const(C::GLV_BETA);
        if constexpr (WIN5) {
          // 27 signed five-bit windows (glv.hpp glv_recode5; the top one is 0 or 1 and costs nothing while the chain has not started):
          // 130 doublings and <= 54 additions where four-bit windows take 132 and <= 66 -- the table is the same 16 entries, now the
          // multiples 1..16, and a negative digit negates the entry's Y like a negative half does.
          if constexpr (C::B3_SMALL != 0) {
            uint32_t pk1[7], pk2[7];
            glv_recode5(k1, pk1);
            glv_recode5(k2, pk2);
            for (int d = 26; d >= 0; d--) {
              const uint32_t b1 = (pk1[d >> 2] >> ((d & 3) * 8)) & 0xFFu, b2 = (pk2[d >> 2] >> ((d & 3) * 8)) & 0xFFu;
              if (started) {
                for (int q = 0; q < 5; q++)
                  r = EcDblSmallB<C>::dbl_quad(r, role);
              }
              if (b1 & 31u) {
                Proj t = tab[(b1 & 31u) - 1];
                if (n1 != ((b1 & 0x80u) != 0)) t.y = F::template neg<4>(F::below4(t.y));
                r = started ? ADD(r, t) : t;
                started = true;
              }
              if (b2 & 31u) {
                Proj t = tab[(b2 & 31u) - 1];
                t.x = F::mul(t.x, beta);
                if (n2 != ((b2 & 0x80u) != 0)) t.y = F::template neg<4>(F::below4(t.y));
                r = started ? ADD(r, t) : t;
                started = true;
              }
            }
          }
          return r;
        }
        for (int d = 32; d >= 0; d--) {
          const uint32_t d1 = (k1[d >> 3] >> ((d & 7) * 4)) & 15u, d2 = (k2[d >> 3] >> ((d & 7) * 4)) & 15u;
          if (started) quad_dbl4(r);
          if (d1) {
            Proj t = tab[d1];
            if (n1) t.y = F::template neg<4>(F::below4(t.y));
            r = started ? ADD(r, t) : t;
            started = true;
          }
          if (d2) {
            Proj t = tab[d2];
            t.x = F::mul(t.x, beta);
            if (n2) t.y = F::template neg<4>(F::below4(t.y));
            r = started ? ADD(r, t) : t;
            started = true;
          }
        }
      }
#else
      for (int d = 63; d >= 0; d--) {
        const uint32_t dig = (k[d >> 3] >> ((d & 7) * 4)) & 15u;
        if (started) {
          quad_dbl4(r);
          if (dig) r = ADD(r, tab[dig]);
        } else if (dig) {
          r = tab[dig];
          started = true;
        }
      }
#endif
      return r;
    }
    // scalar given as packed Montgomery words (twiddle / coset tables) -> canonical words
    static __device__ void canonical_from_mont(uint32_t* out, const uint32_t* __restrict__ mont_words)
    {
      uint32_t w[8];
      load8(w, mont_words);
      FR::to_canonical(out, FR::unpack(w));
    }
    static __device__ Proj neg(const Proj& p)
    {
      Proj r = p;
      r.y = F::template neg<4>(p.y);
      return r;
    }
  };

  struct EcLayout {
    uint64_t n;
    uint32_t logn, batch;
    uint64_t bs, es; // point index of logical memory slot m of transform b: b*bs + m*es
    int in_rev, out_rev, inverse, coset;
    uint32_t log_max;
  };

  // work[b][i] = P_j with j = bitrev(i) (DIT wants its input bit-reversed); the forward coset factor g^j follows in k_ecntt_scale
  template <class C>
  __global__ __launch_bounds__(64) void k_ecntt_load(const uint32_t* __restrict__ in, typename EC<C>::Proj* __restrict__ work, EcLayout lay)
  {
    using T = EcNtt<C>;
    using E = typename T::E;
    using F = typename T::F;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= lay.n * lay.batch) return;
    const uint64_t b = t / lay.n, i = t % lay.n;
    const uint64_t j = bitrev64(i, lay.logn);
    const uint64_t m = lay.in_rev ? i : j; // memory slot of logical x[j]
    const uint32_t* w = in + (b * lay.bs + m * lay.es) * 3 * E::N32;
    typename E::Proj p;
    p.x = F::from_canonical(w);
    p.y = F::from_canonical(w + E::N32);
    p.z = F::from_canonical(w + 2 * E::N32);
    work[b * lay.n + i] = p; // (the coset factor g^j is applied by k_ecntt_scale)
  }

  // work[b][i] *= s_i, FOUR lanes per point sharing the doubling chain and the additions (mul_words_quad) -- the factors a transform
  // applies to every point outside its butterflies: g^j on the way in (forward coset; i = bit-reversed j), 1/N and g^-k on the way
  // out (inverse), folded into ONE scalar per point. Rounds 2-3 did this bit-serially on one lane inside the load / store kernels:
  // 1.8 ms (1/N) to 4.8 ms (coset + 1/N) on top of a 2.5 ms transform of 2^10 points; the quad chain is ~1.0 ms.
  // mode 1: s_i = coset_pow[bitrev(i)] ; mode 2: s_k = ninv [* coset_pow[k]]. Tables: LDS, or global when gtabs != nullptr.
  template <class C>
  __global__ __launch_bounds__(64) void k_ecntt_scale(typename EC<C>::Proj* __restrict__ work, const uint32_t* __restrict__ coset_pow, BigWords ninv_mont, EcLayout lay, int mode, typename EC<C>::Proj* __restrict__ gtabs)
  {
    using T = EcNtt<C>;
    using E = typename T::E;
    using FR = typename T::FR;
    extern __shared__ uint32_t scale_tabs_raw[];
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint32_t role = threadIdx.x & 3u;
    const uint64_t npts = lay.n * lay.batch;
    const bool live = (lane >> 2) < npts;
    const uint64_t t = live ? (lane >> 2) : npts - 1; // (a quad past the end redoes the last point and stores nothing)
    const uint64_t i = t % lay.n;
    typename E::Proj* tab = gtabs ? gtabs + (size_t)(lane >> 2) * 16 : reinterpret_cast<typename E::Proj*>(scale_tabs_raw) + (size_t)(threadIdx.x >> 2) * 16;
    uint32_t w[8], k[8];
    typename FR::fe sc;
    if (mode == 1) {
      load8(w, coset_pow + bitrev64(i, lay.logn) * 8);
      sc = FR::unpack(w);
    } else {
      sc = FR::unpack(ninv_mont.w);
      if (lay.coset) {
        load8(w, coset_pow + i * 8);
        sc = FR::mul(sc, FR::unpack(w));
      }
    }
    FR::to_canonical(k, sc);
    // (a quad past the end works on the identity, not on the slot the last live quad is overwriting in place: ADVICE r04)
    const typename E::Proj pin = live ? work[t] : E::proj_identity();
    const typename E::Proj p = T::mul_words_quad(pin, k, role, tab);
    if (live && role == 0) work[t] = p;
  }

  // stage q: pairs (i, i + 2^q) inside blocks of 2^(q+1); twiddle w_n^(pos * n / 2^(q+1)).
  // Four lanes (one DPP quad) per butterfly: see EcNtt::mul_words<QUAD>. All four lanes load the same pair, lane 0 stores.
  // gtabs != nullptr: the quads' 16-entry tables live in global memory (L2-resident: one 108 / 168-byte entry is read per
  // window, ~6 % of a chain's latency) instead of LDS. LDS holds 16 tables per 64-thread block = 27.6 / 43 KB, i.e. 5 / 3 waves
  // per CU = 20480 / 12288 butterflies in flight; a stage with more than that (2^16 points: 32768) took three rounds of
  // ~1 ms. With the tables out of LDS the registers bound the occupancy (2 waves per SIMD = 32768 quads): 2^16 46.6 -> see
  // profiles/r04_ecntt.txt.
  template <class C>
  __global__ __launch_bounds__(64) void k_ecntt_stage(typename EC<C>::Proj* __restrict__ work, const uint32_t* __restrict__ tw, EcLayout lay, int q, typename EC<C>::Proj* __restrict__ gtabs)
  {
    using T = EcNtt<C>;
    using E = typename T::E;
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint32_t role = threadIdx.x & 3u;
    const uint64_t half_n = lay.n >> 1;
    const uint64_t nbf = half_n * lay.batch;
    // a quad past the end redoes the last butterfly (same values) and does not store: every lane of a wave runs the same
    // trip counts, which the DPP exchanges of the doubling chain rely on
    const bool live = (lane >> 2) < nbf;
    const uint64_t t = live ? (lane >> 2) : nbf - 1;
    const uint64_t b = t / half_n, bf = t % half_n;
    const uint64_t half = (uint64_t)1 << q;
    const uint64_t pos = bf & (half - 1);
    const uint64_t i = ((bf >> q) << (q + 1)) + pos;
    typename E::Proj* base = work + b * lay.n;
    const typename E::Proj u = live ? base[i] : E::proj_identity(); // (dead quads: not the slots the last live quad rewrites in place)
    typename E::Proj v = live ? base[i + half] : E::proj_identity();
#ifdef ECNTT_OLD_MUL
    if (pos != 0) {
      const uint64_t max_mask = ((uint64_t)1 << lay.log_max) - 1;
      uint64_t idx = (pos << (lay.logn - 1 - q)) << (lay.log_max - lay.logn);
      if (lay.inverse) idx = (((uint64_t)1 << lay.log_max) - idx) & max_mask;
      uint32_t k[8];
      T::canonical_from_mont(k, tw + idx * 8);
      v = T::mul_words_serial(v, k);
    }
#else
    {
      extern __shared__ uint32_t stage_tabs_raw[]; // [quad][multiple] when the tables are in LDS (dynamic size: 0 otherwise)
      typename E::Proj* tab = gtabs ? gtabs + (size_t)(lane >> 2) * 16 : reinterpret_cast<typename E::Proj*>(stage_tabs_raw) + (size_t)(threadIdx.x >> 2) * 16;
      const uint64_t max_mask = ((uint64_t)1 << lay.log_max) - 1;
      uint64_t idx = (pos << (lay.logn - 1 - q)) << (lay.log_max - lay.logn);
      if (lay.inverse) idx = (((uint64_t)1 << lay.log_max) - idx) & max_mask;
      uint32_t k[8];
      T::canonical_from_mont(k, tw + (pos != 0 ? idx : 0) * 8);
      if (pos == 0) { // w^0 = 1: k = 1, the multiplication returns v itself (one table read, no doubling)
#pragma unroll
        for (int w = 0; w < 8; w++)
          k[w] = w == 0 ? 1u : 0u;
      }
      v = T::mul_words_quad(v, k, role, tab);
    }
#endif
#ifdef ECNTT_NOQUADADD
    if (live && role == 0) {
      base[i] = E::add(u, v);
      base[i + half] = E::add(u, T::neg(v));
    }
#else
    const typename E::Proj s0 = QADD(u, v, role), s1 = QADD(u, T::neg(v), role);
    if (live && role == 0) {
      base[i] = s0;
      base[i + half] = s1;
    }
#endif
  }

  // ---- radix-2^r "matrix form" stages: r radix-2 stages for the LATENCY of one scalar multiplication -----------------
  // A stage's time is the latency of one k * P chain (~1 ms), whatever the number of butterflies, as long as they fit the
  // chip -- and a transform of 2^10 .. 2^13 points has far fewer butterflies than the chip has lanes. So r stages are taken
  // at once: with L = 2^q0, R = 2^r, M = L R, the R sub-transforms A_j (memory order j, i.e. residue rev_r(j) mod R)
  // combine as            Y[pos + L k] = sum_j  w_M^(rev(j) (pos + L k)) A_j[pos],        k = 0 .. R-1,
  // and EVERY product of that sum is computed at the same time, one DPP quad each (k_ecntt_terms), followed by a
  // short chain of additions (k_ecntt_sums). Only u = k mod R/2 needs a product of its own: w_M^(rev(j) L R/2) = (-1)^rev(j),
  // so Y[u + R/2] takes the terms of the odd residues (j >= R/2) with a minus sign. n (R - 1) / 2 scalar multiplications per
  // stage instead of n / 2 -- (R - 1) times the work for 1 / r of the latency: 2^10 points run as 5 + 5 (2 chains instead of
  // 10), 2^12 as 3 + 3 + 3 + 3; where the work would no longer fit one round of quads (2^14 and up) r falls back to 1.
  // The R / 2 quads that multiply the same A_j share its 16-entry table in LDS (they build identical copies of it).
  struct EcStage {
    int q0, r; // covers the radix-2 stages q0 .. q0 + r - 1
  };
  template <class C>
  __global__ __launch_bounds__(64) void k_ecntt_terms(const typename EC<C>::Proj* __restrict__ work, typename EC<C>::Proj* __restrict__ terms, const uint32_t* __restrict__ tw, EcLayout lay, EcStage sg)
  {
    using T = EcNtt<C>;
    using E = typename T::E;
    extern __shared__ uint32_t tabs_raw[]; // [16 / hr tables][16 multiples]: quads with the same (group, j) share a table
    typename E::Proj* tabs = reinterpret_cast<typename E::Proj*>(tabs_raw);
    const uint32_t role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
    const uint32_t R = 1u << sg.r, hr = R >> 1;         // hr = products per (group, j)
    const uint64_t L = (uint64_t)1 << sg.q0;
    const uint64_t ngroups = (lay.n >> sg.r) * lay.batch; // (transform, block of M, pos)
    const uint64_t nitems = ngroups * (R - 1) * hr;
    const uint64_t item0 = (blockIdx.x * (uint64_t)16 + quad);
    const bool live = item0 < nitems;
    const uint64_t item = live ? item0 : nitems - 1; // (a quad past the end redoes the last item and stores nothing)
    const uint32_t u = (uint32_t)(item % hr);
    const uint64_t pair = item / hr;
    const uint32_t j = (uint32_t)(pair % (R - 1)) + 1;
    const uint64_t grp = pair / (R - 1);
    const uint64_t per_t = lay.n >> sg.r; // groups per transform
    const uint64_t b = grp / per_t, g = grp % per_t;
    const uint64_t pos = g & (L - 1), blk = g >> sg.q0;
    const typename E::Proj a = work[b * lay.n + (blk << (sg.q0 + sg.r)) + (uint64_t)j * L + pos];
    // exponent rev_r(j) * (pos + L u) mod M, as an index into the domain table (w_max^i)
    const uint64_t e = ecntt_term_exponent(sg.q0, sg.r, j, u, pos);
    const uint64_t max_mask = ((uint64_t)1 << lay.log_max) - 1;
    uint64_t idx = e << (lay.log_max - (uint32_t)(sg.q0 + sg.r));
    if (lay.inverse) idx = (((uint64_t)1 << lay.log_max) - idx) & max_mask;
    uint32_t k[8];
    T::canonical_from_mont(k, tw + idx * 8); // (w^0 = 1 is in the table: k = 1, one table read and no doubling)
    const typename E::Proj t = T::mul_words_quad(a, k, role, tabs + (size_t)(quad / hr) * 16);
    if (live && role == 0) terms[item] = t;
  }
  // Y[u] = A_0 + sum_{j < R/2} T_j + sum_{j >= R/2} T_j ,  Y[u + R/2] = A_0 + sum_{j < R/2} T_j - sum_{j >= R/2} T_j ; one quad per pair
  template <class C>
  __global__ __launch_bounds__(64) void k_ecntt_sums(const typename EC<C>::Proj* __restrict__ work, const typename EC<C>::Proj* __restrict__ terms, typename EC<C>::Proj* __restrict__ next, EcLayout lay, EcStage sg)
  {
    using T = EcNtt<C>;
    using E = typename T::E;
    const uint32_t role = threadIdx.x & 3u;
    const uint32_t R = 1u << sg.r, hr = R >> 1;
    const uint64_t L = (uint64_t)1 << sg.q0;
    const uint64_t per_t = lay.n >> sg.r;
    const uint64_t ngroups = per_t * lay.batch;
    const uint64_t nout = ngroups * hr;
    const uint64_t o0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 2;
    const bool live = o0 < nout;
    const uint64_t o = live ? o0 : nout - 1;
    const uint32_t u = (uint32_t)(o % hr);
    const uint64_t grp = o / hr;
    const uint64_t b = grp / per_t, g = grp % per_t;
    const uint64_t pos = g & (L - 1), blk = g >> sg.q0;
    const uint64_t base = b * lay.n + (blk << (sg.q0 + sg.r)) + pos;
    const typename E::Proj* tg = terms + (grp * (R - 1)) * hr + u; // term (j, u) at tg[(j - 1) * hr]
    typename E::Proj ev = work[base], od = tg[(size_t)(hr - 1) * hr]; // A_0 ; T_{R/2}
    for (uint32_t j = 1; j < hr; j++)
      ev = QADD(ev, tg[(size_t)(j - 1) * hr], role);
    for (uint32_t j = hr + 1; j < R; j++)
      od = QADD(od, tg[(size_t)(j - 1) * hr], role);
    const typename E::Proj y0 = QADD(ev, od, role), y1 = QADD(ev, T::neg(od), role);
    if (live && role == 0) {
      next[base + (uint64_t)u * L] = y0;
      next[base + (uint64_t)(u + hr) * L] = y1;
    }
  }

  // out[slot(k)] = work[b][k] in the reference's projective_t layout (1/N and g^-k have been applied by k_ecntt_scale)
  template <class C>
  __global__ __launch_bounds__(64) void k_ecntt_store(const typename EC<C>::Proj* __restrict__ work, uint32_t* __restrict__ out, EcLayout lay)
  {
    using T = EcNtt<C>;
    using E = typename T::E;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= lay.n * lay.batch) return;
    const uint64_t b = t / lay.n, k = t % lay.n;
    const typename E::Proj p = work[b * lay.n + k]; // (1/N and g^-k were applied by k_ecntt_scale)
    const uint64_t m = lay.out_rev ? bitrev64(k, lay.logn) : k;
    E::store_proj_canonical(out + (b * lay.bs + m * lay.es) * 3 * E::N32, p);
  }

  template <class C>
  static icicle_error_t ecntt_run(const void* input_v, int size, int dir, const icicle_ntt_config_u256_t* cfg, void* output_v)
  {
    using PR = typename C::fr;
    using FR = FieldOps<PR>;
    using E = EC<C>;
    using Proj = typename E::Proj;
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (size <= 0 || (size & (size - 1)) != 0) return ICICLE_INVALID_ARGUMENT; // cpu_ntt_main.h:38-41
    if (!input_v || !output_v) return ICICLE_INVALID_POINTER;
    if (dir != ICICLE_NTT_FORWARD && dir != ICICLE_NTT_INVERSE) return ICICLE_INVALID_ARGUMENT;
    if (cfg->ordering < 0 || cfg->ordering > ICICLE_kMN) return ICICLE_INVALID_ARGUMENT;
    const int batch = std::max(1, cfg->batch_size);
    ICICLE_TRY(bind_current_device());
    BigDomain dom;
    {
      std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
      auto it = BigDomainStore<PR>::map().find(current_device_id());
      if (it == BigDomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT; // domain not initialised
      dom = it->second;
    }
    int logn = 0;
    while ((1 << logn) < size)
      logn++;
    if (logn > dom.log_max) return ICICLE_INVALID_ARGUMENT;
    if (words_is_zero(cfg->coset_gen) || !words_lt_p<PR>(cfg->coset_gen)) return ICICLE_INVALID_ARGUMENT;

    hipStream_t st = (hipStream_t)cfg->stream;
    const uint64_t n = (uint64_t)size;
    constexpr size_t PWB = (size_t)3 * E::N32 * 4; // bytes per projective_t
    const size_t bytes = (size_t)n * batch * PWB;

    TempBuf d_in_tmp, d_out_tmp, d_pw, d_work, d_terms, d_next; // (d_terms / d_next: the matrix-form stages; alive until the store kernel is queued)
    const uint32_t* d_in = (const uint32_t*)input_v;
    uint32_t* d_out = (uint32_t*)output_v;
    if (!cfg->are_inputs_on_device) {
      HIP_TRY(d_in_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_in_tmp.ptr(), input_v, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_in = d_in_tmp.as<uint32_t>();
    }
    if (!cfg->are_outputs_on_device) {
      HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      d_out = d_out_tmp.as<uint32_t>();
    }
    HIP_TRY(d_work.alloc((size_t)n * batch * sizeof(Proj), st), ICICLE_ALLOCATION_FAILED);

    EcLayout lay;
    lay.n = n;
    lay.logn = (uint32_t)logn;
    lay.batch = (uint32_t)batch;
    if (cfg->columns_batch) { // element j of transform b at j*batch + b (ntt_cpu.h:250,274-275)
      lay.bs = 1;
      lay.es = (uint64_t)batch;
    } else {
      lay.bs = n;
      lay.es = 1;
    }
    const int ord = cfg->ordering;
    lay.in_rev = (ord == ICICLE_kRN || ord == ICICLE_kRR);
    lay.out_rev = (ord == ICICLE_kNR || ord == ICICLE_kRR);
    lay.inverse = (dir == ICICLE_NTT_INVERSE);
    lay.coset = !words_is_one(cfg->coset_gen);
    lay.log_max = (uint32_t)dom.log_max;

    if (lay.coset) {
      HIP_TRY(d_pw.alloc(n * 32, st), ICICLE_ALLOCATION_FAILED);
      typename FR::fe gm = FR::from_canonical(cfg->coset_gen);
      if (lay.inverse) gm = host_inverse<PR>(gm);
      k_big_coset_powers<PR><<<(unsigned)((n / 16 + 256) / 256), 256, 0, st>>>(d_pw.as<uint32_t>(), mont_words<PR>(gm), n);
      LAUNCH_CHECK("k_big_coset_powers", st);
    }
    BigWords ninv_mont{};
    if (lay.inverse) ninv_mont = mont_words<PR>(host_ninv<PR>(logn)); // packed Montgomery words, like the coset table's entries

    const uint64_t tot = n * batch;
    Proj* work = d_work.as<Proj>();
    k_ecntt_load<C><<<(unsigned)((tot + 63) / 64), 64, 0, st>>>(d_in, work, lay);
    LAUNCH_CHECK("k_ecntt_load", st);
    // per-point factors outside the butterflies (forward coset on the way in; 1/N [and the coset] on the way out): k_ecntt_scale
    const uint64_t scale_quads = ((tot * 4 + 63) / 64) * 16;
    const bool scale_gtab = tot > (uint64_t)(sizeof(Proj) > 120 ? 12288 : 20480); // more points than LDS-resident tables allow in flight
    // ONE scratch buffer (d_terms) serves the scale tables, the stage tables and the matrix-form terms: they are never alive
    // at the same time on the stream (rounds 1-4 kept the scale tables next to the stage scratch: 2^20 BLS12-381 points took
    // 2.4 GB + 1.2 GB for a 151 MB transform). Sized once, for the largest of the three.
    static const int forced_r = getenv("ICICLE_HIP_ECNTT_RADIX_LOG") ? atoi(getenv("ICICLE_HIP_ECNTT_RADIX_LOG")) : 0;
    // product budget of a matrix-form stage: 16384 quads is the knee for transforms of up to 2^13 points (2^12: 2.04 ms against 2.71 at 32768),
    // from 2^14 points a second radix-4 stage pays (2^14: 6.49 -> 6.02 ms); profiles/r06_ecntt_quads_budget.txt, re-measured on the round-6 chain
    static const uint64_t budget_env = getenv("ICICLE_HIP_ECNTT_QUADS") ? (uint64_t)atoll(getenv("ICICLE_HIP_ECNTT_QUADS")) : 0;
    const uint64_t budget = budget_env ? budget_env : (tot >= 16384 ? 32768 : 16384);
    static const int gtab_mode = getenv("ICICLE_HIP_ECNTT_GTABS") ? atoi(getenv("ICICLE_HIP_ECNTT_GTABS")) : -1; // 0 / 1 force, -1 auto
    int widths[64];
    const int nst = ecntt_stage_plan(logn, tot, budget, forced_r, widths);
    const int rmax = nst ? widths[0] : 1; // (widths are non-increasing)
    const uint64_t stage_quads = ((tot / 2 * 4 + 63) / 64) * 16; // quads launched per radix-2 stage (incl. the padding of the last block)
    const bool stage_gtab = gtab_mode >= 0 ? gtab_mode != 0 : tot / 2 > (uint64_t)(sizeof(Proj) > 120 ? 12288 : 20480);
    {
      size_t need = 0;
      if (scale_gtab && (lay.inverse || lay.coset)) need = std::max(need, (size_t)scale_quads * 16 * sizeof(Proj));
      if (rmax == 1 && stage_gtab) need = std::max(need, (size_t)stage_quads * 16 * sizeof(Proj));
      if (rmax > 1 && logn > 0) need = std::max(need, (size_t)(tot * ((1ull << rmax) - 1) / 2) * sizeof(Proj)); // n (R - 1) / 2 products of the widest stage
      if (need) HIP_TRY(d_terms.alloc(need, st), ICICLE_ALLOCATION_FAILED);
    }
    auto scale = [&](Proj* pts, int mode) -> icicle_error_t {
      Proj* gt = scale_gtab ? d_terms.as<Proj>() : nullptr;
      k_ecntt_scale<C><<<(unsigned)((tot * 4 + 63) / 64), 64, scale_gtab ? 0 : (size_t)16 * 16 * sizeof(Proj), st>>>(pts, d_pw.as<uint32_t>(), ninv_mont, lay, mode, gt);
      LAUNCH_CHECK("k_ecntt_scale", st);
      return ICICLE_SUCCESS;
    };
    if (lay.coset && !lay.inverse) ICICLE_TRY(scale(work, 1));
    // stage radix (planned above): the largest r <= 5 whose n (R - 1) / 2 products still fit one round of quads on the chip (the
    // budget is a measured knee, not a hard limit: beyond it a stage simply takes a second round); ICICLE_HIP_ECNTT_RADIX_LOG forces r
    if (rmax == 1) {
      // more butterflies per stage than LDS-resident tables allow in flight: tables in global memory (see k_ecntt_stage)
      const bool gtab = stage_gtab;
      Proj* gt = gtab ? d_terms.as<Proj>() : nullptr;
      for (int q = 0; q < logn; q++) {
        k_ecntt_stage<C><<<(unsigned)((tot / 2 * 4 + 63) / 64), 64, gtab ? 0 : (size_t)16 * 16 * sizeof(Proj), st>>>(work, dom.tw, lay, q, gt);
        LAUNCH_CHECK("k_ecntt_stage", st);
      }
    } else if (logn > 0) {
      HIP_TRY(d_next.alloc((size_t)tot * sizeof(Proj), st), ICICLE_ALLOCATION_FAILED);
      Proj* cur = work;
      Proj* nxt = d_next.as<Proj>();
      int q0 = 0;
      for (int si = 0; si < nst; si++) {
        const int r = widths[si];
        const EcStage sg{q0, r};
        const uint64_t items = (tot >> r) * ((1ull << r) - 1) * (1ull << (r - 1));
        const size_t tab_bytes = (size_t)std::max(1, 16 >> (r - 1)) * 16 * sizeof(Proj); // 16 / (R / 2) tables per 16-quad block
        k_ecntt_terms<C><<<(unsigned)((items + 15) / 16), 64, tab_bytes, st>>>(cur, d_terms.as<Proj>(), dom.tw, lay, sg);
        LAUNCH_CHECK("k_ecntt_terms", st);
        const uint64_t nout = tot >> 1;
        k_ecntt_sums<C><<<(unsigned)((nout * 4 + 63) / 64), 64, 0, st>>>(cur, d_terms.as<Proj>(), nxt, lay, sg);
        LAUNCH_CHECK("k_ecntt_sums", st);
        std::swap(cur, nxt);
        q0 += r;
      }
      work = cur;
    }
    if (lay.inverse) ICICLE_TRY(scale(work, 2));
    k_ecntt_store<C><<<(unsigned)((tot + 63) / 64), 64, 0, st>>>(work, d_out, lay);
    LAUNCH_CHECK("k_ecntt_store", st);
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);

    if (!cfg->are_outputs_on_device) {
      HIP_TRY(hipMemcpyAsync(output_v, d_out, bytes, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    } else if (!cfg->is_async) {
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    }
    return ICICLE_SUCCESS;
  }

} // namespace icicle_hip

using namespace icicle_hip;

#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

#define DEFINE_ECNTT(C)                                                                                                \
  extern "C" icicle_error_t C##_ecntt(const void* input, int size, int dir, const icicle_ntt_config_u256_t* config, void* output) \
  {                                                                                                                    \
    GUARDED(ecntt_run<C##_g1>(input, size, dir, config, output));                                                      \
  }                                                                                                                    \
  extern "C" icicle_error_t icicle_hip_##C##_ecntt(const void* i, int n, int d, const icicle_ntt_config_u256_t* c, void* o) { GUARDED(ecntt_run<C##_g1>(i, n, d, c, o)); }

DEFINE_ECNTT(bn254)
DEFINE_ECNTT(bls12_381)
DEFINE_ECNTT(bls12_377)
