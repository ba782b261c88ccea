// TEST-ONLY host build of the device arithmetic headers (bigfield.hpp / ec.hpp / smallfield.hpp)
// with the debug bound tracker on (-DBIGFIELD_BOUNDS). Lets the CPU test-suite check the exact
// code the kernels run against Python big-int arithmetic and the reference oracle, without a GPU.
// Not part of the shipped library; nothing in icicle_amd/ links it.
#include <cstdint>
#include <cstring>
#include "../icicle_amd/csrc/ec.hpp"
#include "../icicle_amd/csrc/smallfield.hpp"
#include "../icicle_amd/csrc/goldfield.hpp"
#include "../icicle_amd/csrc/glv.hpp"
#include "../icicle_amd/csrc/ec_dbl_quad.hpp"

using namespace icicle_hip;

namespace {
  template <class PR>
  int field_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out)
  {
    using F = FieldOps<PR>;
    typename F::fe x = F::from_canonical(a), y = F::from_canonical(b), r;
    switch (op) {
    case 0: r = F::mul(x, y); break;
    case 1: r = F::sqr(x); break;
    case 2: r = F::add(x, y); break;
    case 3: r = F::template sub<2>(x, y); break;
    case 4: r = F::template neg<2>(x); break;
    case 5: { // stress lazy bounds: ((x+y)+(x+y)) * (x - y + 8p) ...
      auto s = F::add(F::add(x, y), F::add(x, y));
      auto d = F::template sub<8>(x, F::add(F::add(y, y), F::add(y, y)));
      r = F::mul(s, d); // 2(x+y)(x-4y)
      break;
    }
    case 6: { // from reference-Montgomery words -> canonical
      r = F::from_refmont(a);
      break;
    }
    case 7: { // canonical -> reference-Montgomery words
      F::to_refmont(out, x);
      return 0;
    }
    case 8: { // is_zero(x - y)
      out[0] = F::is_zero(F::template sub<2>(x, y)) ? 1 : 0;
      return 0;
    }
    case 9: r = F::inv(x); break; // x^(p-2); 0 -> 0 (the reference's inverse(0) = 0, projective.h:55-59)
    default: return -1;
    }
    F::to_canonical(out, r);
    return 0;
  }

  // points: affine canonical words (x,y); identity (0,0)
  template <class C>
  int ec_op(int op, const uint32_t* pts, int n, const uint32_t* aux, uint32_t* out)
  {
    using E = EC<C>;
    using F = typename E::F;
    constexpr int N32 = E::N32;
    // Montgomery-form affine point (cold kernels: precompute, generator, complete adds)
    auto load = [&](const uint32_t* w) {
      typename E::Aff a;
      a.x = F::from_canonical(w);
      a.y = F::from_canonical(w + N32);
      return a;
    };
    switch (op) {
    case 0: { // XYZZ accumulate all points (aux[i]&1 = negate), output projective canonical
      typename E::XYZZ acc;
      bool empty = true;
      for (int i = 0; i < n; i++) {
        const uint32_t* w = pts + (size_t)i * 2 * N32;
        if (E::words_are_zero(w)) continue;
        // the hot loop consumes the canonical words exactly as they lie in HBM (ec.hpp scaling convention)
        auto a = E::cneg(E::load_plain(w), aux && (aux[i] & 1));
        E::madd(acc, empty, a);
      }
      E::store_proj_canonical(out, E::to_proj(acc, empty));
      return 0;
    }
    case 1: { // complete projective sum of all points (identity allowed)
      auto acc = E::proj_identity();
      for (int i = 0; i < n; i++) {
        const uint32_t* w = pts + (size_t)i * 2 * N32;
        if (E::words_are_zero(w)) {
          acc = E::add(acc, E::proj_identity());
          continue;
        }
        acc = E::add(acc, E::to_proj(E::cneg(load(w), aux && (aux[i] & 1))));
      }
      E::store_proj_canonical(out, acc);
      return 0;
    }
    case 2: { // mul_small: aux[0] * pts[0]
      auto p = E::words_are_zero(pts) ? E::proj_identity() : E::to_proj(load(pts));
      E::store_proj_canonical(out, E::mul_small(p, aux[0]));
      return 0;
    }
    case 3: { // generator
      E::store_proj_canonical(out, E::to_proj(E::generator()));
      return 0;
    }
    case 4: { // repeated doubling: 2^aux[0] * pts[0] via complete dbl
      auto p = E::to_proj(load(pts));
      for (uint32_t i = 0; i < aux[0]; i++)
        p = E::dbl(p);
      E::store_proj_canonical(out, p);
      return 0;
    }
    case 5: { // the window-combine chain: 2^aux[0] * pts[0] via to_jac / dbl_jac / from_jac
      auto p = E::words_are_zero(pts) ? E::proj_identity() : E::add(E::to_proj(load(pts)), E::proj_identity()); // a non-trivial Z
      auto j = E::to_jac(p);
      for (uint32_t i = 0; i < aux[0]; i++)
        j = E::dbl_jac(j);
      E::store_proj_canonical(out, E::from_jac(j));
      return 0;
    }
    case 6: { // msm_precompute_bases' chain: 2^aux[0] * pts[0] via dbl_jac_lazy from Z = 1, reduced at the end (bounds tracked)
      if (E::words_are_zero(pts)) {
        E::store_proj_canonical(out, E::proj_identity());
        return 0;
      }
      typename E::Jac j;
      const auto a = load(pts);
      j.x = a.x, j.y = a.y, j.z = F::one();
      for (uint32_t i = 0; i < aux[0]; i++)
        j = E::dbl_jac_lazy(j);
      j.x = F::reduce(j.x), j.y = F::reduce(j.y), j.z = F::reduce(j.z);
      E::store_proj_canonical(out, E::from_jac(j));
      return 0;
    }
    case 7: { // the ECNTT butterflies' scalar multiplication (ecntt.hip mul_words_quad, one lane's arithmetic): aux[0..7] * pts[0] by the GLV
              // split, 33 joint four-bit windows, four lazily reduced doublings per window with Y brought back below 4p for the complete addition
      if constexpr (C::EXT_DEGREE == 1) {
        using Proj = typename E::Proj;
        const Proj p = E::words_are_zero(pts) ? E::proj_identity() : E::to_proj(load(pts));
        Proj tab[16];
        uint32_t k1[5], k2[5];
        bool n1, n2;
        glv_decompose<C>(aux, k1, n1, k2, n2);
        const typename F::fe beta = F::from_const(C::GLV_BETA);
        Proj r = E::proj_identity();
        bool started = false;
        if constexpr (C::B3_SMALL != 0) { // signed five-bit windows over the multiples 1..16, projective doublings (the device's WIN5 path)
          Proj e = p;
          for (int i = 0; i < 16; i++) {
            tab[i] = e;
            if (i < 15) e = (i == 0) ? EcDblSmallB<C>::dbl(p) : E::add(e, p);
          }
          uint32_t pk1[7], pk2[7];
          glv_recode5(k1, pk1);
          glv_recode5(k2, pk2);
          for (int d = 26; d >= 0; d--) {
            const uint32_t b1 = (pk1[d >> 2] >> ((d & 3) * 8)) & 0xFFu, b2 = (pk2[d >> 2] >> ((d & 3) * 8)) & 0xFFu;
            if (started)
              for (int q = 0; q < 5; q++)
                r = EcDblSmallB<C>::dbl(r);
            if (b1 & 31u) {
              Proj t = tab[(b1 & 31u) - 1];
              if (n1 != ((b1 & 0x80u) != 0)) t.y = F::template neg<4>(F::below4(t.y));
              r = started ? E::add(r, t) : t;
              started = true;
            }
            if (b2 & 31u) {
              Proj t = tab[(b2 & 31u) - 1];
              t.x = F::mul(t.x, beta);
              if (n2 != ((b2 & 0x80u) != 0)) t.y = F::template neg<4>(F::below4(t.y));
              r = started ? E::add(r, t) : t;
              started = true;
            }
          }
          Proj nr = r;
          nr.y = F::template neg<4>(r.y);
          E::store_proj_canonical(out, E::add(E::add(r, nr), r));
          return 0;
        }
        Proj e = E::proj_identity();
        for (int i = 0; i < 16; i++) {
          tab[i] = e;
          e = (i == 0) ? p : ((i == 1) ? E::dbl(p) : E::add(e, p));
        }
        for (int d = 32; d >= 0; d--) {
          const uint32_t d1 = (k1[d >> 3] >> ((d & 7) * 4)) & 15u, d2 = (k2[d >> 3] >> ((d & 7) * 4)) & 15u;
          if (started) {
            if constexpr (C::B3_SMALL != 0) {
              for (int q = 0; q < 4; q++)
                r = EcDblSmallB<C>::dbl(r); // ec_dbl_quad.hpp: the quad form's operand flow on one lane
            } else {
              typename E::Jac j = E::to_jac(r);
              for (int q = 0; q < 4; q++)
                j = E::dbl_jac_lazy(j);
              F::template cond_sub<16>(j.y);
              j.y = F::below4(j.y);
              r = E::from_jac(j);
            }
          }
          if (d1) {
            Proj t = tab[d1];
            if (n1) t.y = F::template neg<4>(F::below4(t.y));
            r = started ? E::add(r, t) : t;
            started = true;
          }
          if (d2) {
            Proj t = tab[d2];
            t.x = F::mul(t.x, beta);
            if (n2) t.y = F::template neg<4>(F::below4(t.y));
            r = started ? E::add(r, t) : t;
            started = true;
          }
        }
        // (the butterfly then forms u + v and u - v: the negation of the result must be in bounds as well)
        Proj nr = r;
        nr.y = F::template neg<4>(r.y);
        E::store_proj_canonical(out, E::add(E::add(r, nr), r));
        return 0;
      }
      return -1;
    }
    default: return -1;
    }
  }
} // namespace

// GLV decomposition (glv.hpp): out = |k1| (5 words), neg1, |k2| (5 words), neg2
extern "C" int host_glv_decompose(int curve, const uint32_t* k, uint32_t* out)
{
  bool n1 = false, n2 = false;
  switch (curve) {
  case 0: glv_decompose<bn254_g1>(k, out, n1, out + 6, n2); break;
  case 1: glv_decompose<bls12_381_g1>(k, out, n1, out + 6, n2); break;
  case 4: glv_decompose<bls12_377_g1>(k, out, n1, out + 6, n2); break;
  case 5: glv_decompose<grumpkin_g1>(k, out, n1, out + 6, n2); break;
  default: return -1;
  }
  out[5] = n1, out[11] = n2;
  return 0;
}

// signed five-bit recoding (glv.hpp glv_recode5): k (5 words) -> 27 digits as int32
extern "C" int host_glv_recode5(const uint32_t* k, int32_t* digits)
{
  uint32_t pk[7];
  glv_recode5(k, pk);
  for (int i = 0; i < 27; i++) {
    const uint32_t b = (pk[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
    digits[i] = (b & 0x80u) ? -(int32_t)(b & 31u) : (int32_t)(b & 31u);
    if (b & 0x60u) return -1; // no other bits
  }
  return 0;
}

extern "C" int host_field_op(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out)
{
  switch (field) {
  case 0: return field_op<bn254_fq_params>(op, a, b, out);
  case 1: return field_op<bn254_fr_params>(op, a, b, out);
  case 2: return field_op<bls12_381_fq_params>(op, a, b, out);
  case 3: return field_op<bls12_381_fr_params>(op, a, b, out);
  case 4: return field_op<bls12_377_fq_params>(op, a, b, out);
  case 5: return field_op<bls12_377_fr_params>(op, a, b, out);
  case 6: return field_op<stark252_fr_params>(op, a, b, out);
  case 7: return field_op<goldilocks_params>(op, a, b, out);
  }
  return -1;
}

extern "C" int host_ec_op(int curve, int op, const uint32_t* pts, int n, const uint32_t* aux, uint32_t* out)
{
  switch (curve) {
  case 0: return ec_op<bn254_g1>(op, pts, n, aux, out);
  case 1: return ec_op<bls12_381_g1>(op, pts, n, aux, out);
  case 2: return ec_op<bn254_g2>(op, pts, n, aux, out);
  case 3: return ec_op<bls12_381_g2>(op, pts, n, aux, out);
  case 4: return ec_op<bls12_377_g1>(op, pts, n, aux, out);
  case 5: return ec_op<grumpkin_g1>(op, pts, n, aux, out);
  case 6: return ec_op<bls12_377_g2>(op, pts, n, aux, out);
  }
  return -1;
}

// 31-bit fields
extern "C" int host_small_op(int field, int op, uint32_t a, uint32_t b, uint32_t* out)
{
  auto run = [&](auto tag) {
    using S = SmallField<decltype(tag)>;
    uint32_t x = S::to_mont(a), y = S::to_mont(b), r;
    switch (op) {
    case 0: r = S::mul(x, y); break;
    case 1: r = S::add(x, y); break;
    case 2: r = S::sub(x, y); break;
    case 3: r = S::pow(x, b); break; // x^b (b plain integer)
    case 4: r = S::inv(x); break;
    default: return -1;
    }
    *out = S::from_mont(r);
    return 0;
  };
  switch (field) {
  case 0: return run(babybear_params{});
  case 1: return run(koalabear_params{});
  }
  return -1;
}
