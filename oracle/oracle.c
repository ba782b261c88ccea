/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY. Plain-C, single-threaded restatement of the reference's
 * algorithm for the MSM / NTT hot path (ICICLE CPU backend). Nothing under icicle_amd/ links,
 * imports or executes this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may (as the checker, never as the thing measured or shipped).
 *
 * Parity pinning: the reference tree holds NO golden vectors for MSM or NTT, so this
 * restatement is pinned against (a) the reference itself, built unmodified into oracle/_ref and run
 * here (tests/test_oracle.py compares them on seeded inputs), (b) the committed fixtures under
 * tests/golden/ minted from oracle/_ref by tests/golden/make_golden.py, and (c) the pure-Python
 * big-int definitions in oracle/pyref.py.
 *
 * Each function cites the reference code it follows (paths relative to /root/reference/icicle).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define MAXL 6 /* 64-bit limbs: 4 (bn254 / grumpkin, 253..255-bit scalars) or 6 (bls12_381 / bls12_377 base field) */

typedef struct {
  int nl;          /* 64-bit limbs */
  uint64_t p[MAXL];
} field_t;

/* moduli: include/icicle/fields/snark_fields/{bn254_base.h:8-9, bn254_scalar.h:9-10,
 * bls12_381_base.h:10-11, bls12_381_scalar.h:11-12} */
static const field_t BN254_FQ = {4, {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const field_t BN254_FR = {4, {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const field_t BLS_FQ = {6, {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};
static const field_t BLS_FR = {4, {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
/* bls12_377_base.h:8-9, bls12_377_scalar.h:9-10 */
static const field_t BLS377_FQ = {6, {0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull, 0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull}};
static const field_t BLS377_FR = {4, {0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull}};

typedef struct {
  uint64_t l[MAXL];
} fe_t;

/* ---- storage<N> helpers (include/icicle/math/storage.h:36-48: little-endian limbs) ---- */
static int fe_is_zero(const field_t* F, const fe_t* a)
{
  uint64_t o = 0;
  for (int i = 0; i < F->nl; i++) o |= a->l[i];
  return o == 0;
}
static int cmp_ge(const uint64_t* a, const uint64_t* b, int n)
{
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return 1;
}
static uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int n)
{ /* host_math.h add_sub_limbs: carry chain */
  u128 c = 0;
  for (int i = 0; i < n; i++) {
    c += (u128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int n)
{
  uint64_t borrow = 0;
  for (int i = 0; i < n; i++) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
/* ModArith::operator+ / - / neg (math/modular_arithmetic.h:354-369, 587-597): add then conditionally
 * subtract p; subtract then conditionally add p. */
static void fe_add(const field_t* F, fe_t* r, const fe_t* a, const fe_t* b)
{
  uint64_t t[MAXL];
  uint64_t c = add_n(t, a->l, b->l, F->nl);
  if (c || cmp_ge(t, F->p, F->nl)) sub_n(t, t, F->p, F->nl);
  memcpy(r->l, t, 8 * F->nl);
}
static void fe_sub(const field_t* F, fe_t* r, const fe_t* a, const fe_t* b)
{
  uint64_t t[MAXL];
  if (sub_n(t, a->l, b->l, F->nl)) add_n(t, t, F->p, F->nl);
  memcpy(r->l, t, 8 * F->nl);
}
static void fe_neg(const field_t* F, fe_t* r, const fe_t* a)
{
  fe_t z;
  memset(&z, 0, sizeof z);
  fe_sub(F, r, &z, a);
}
/* ModArith::operator* (modular_arithmetic.h:517-521) = multiply_raw (host_math.h:254-281, schoolbook on
 * u64 limbs via __uint128_t, :227-238) followed by a full reduction mod p. The reference reduces with
 * its multi-precision Barrett (host_math.h:438-470); any exact reduction yields the same canonical
 * residue, here shift-and-subtract long division on the 2N-limb product. */
static void fe_mul(const field_t* F, fe_t* r, const fe_t* a, const fe_t* b)
{
  const int n = F->nl;
  uint64_t t[2 * MAXL + 1];
  memset(t, 0, sizeof t);
  for (int i = 0; i < n; i++) {
    u128 c = 0;
    for (int j = 0; j < n; j++) {
      c += (u128)a->l[i] * b->l[j] + t[i + j];
      t[i + j] = (uint64_t)c;
      c >>= 64;
    }
    t[i + n] = (uint64_t)c;
  }
  /* remainder of t (2n limbs) mod p by bitwise long division */
  uint64_t rem[MAXL + 1];
  memset(rem, 0, sizeof rem);
  for (int bit = 128 * n - 1; bit >= 0; bit--) {
    /* rem = rem*2 + bit */
    uint64_t carry = (t[bit >> 6] >> (bit & 63)) & 1;
    for (int i = 0; i <= n; i++) {
      uint64_t nc = rem[i] >> 63;
      rem[i] = (rem[i] << 1) | carry;
      carry = nc;
    }
    if (rem[n] || cmp_ge(rem, F->p, n)) {
      uint64_t bo = sub_n(rem, rem, F->p, n);
      rem[n] -= bo;
    }
  }
  memcpy(r->l, rem, 8 * n);
}
static void fe_sqr(const field_t* F, fe_t* r, const fe_t* a) { fe_mul(F, r, a, a); }
static void fe_set_u32(const field_t* F, fe_t* r, uint32_t v)
{
  (void)F;
  memset(r, 0, sizeof *r);
  r->l[0] = v;
}
/* ModArith::inverse (modular_arithmetic.h:621-657) is a binary extended GCD with inverse(0) = 0; the
 * value is the same as Fermat a^(p-2), used here. */
static void fe_inv(const field_t* F, fe_t* r, const fe_t* a)
{
  uint64_t e[MAXL];
  uint64_t two[MAXL] = {2};
  sub_n(e, F->p, two, F->nl);
  fe_t acc, base = *a;
  fe_set_u32(F, &acc, 1);
  for (int i = 0; i < 64 * F->nl; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, &base);
    fe_sqr(F, &base, &base);
  }
  *r = acc;
}
static void fe_load(const field_t* F, fe_t* r, const uint32_t* w)
{
  memset(r, 0, sizeof *r);
  for (int i = 0; i < F->nl; i++) r->l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
static void fe_store(const field_t* F, uint32_t* w, const fe_t* a)
{
  for (int i = 0; i < F->nl; i++) {
    w[2 * i] = (uint32_t)a->l[i];
    w[2 * i + 1] = (uint32_t)(a->l[i] >> 32);
  }
}

/* ===================================================================================== curves */
typedef struct {
  const field_t* fq;
  const field_t* fr;
  uint32_t b; /* |weierstrass_b|: 3 (curves/params/bn254.h:25), 4 (params/bls12_381.h:26), 1 (params/bls12_377.h), 17 (params/grumpkin.h) */
  int b_neg;  /* is_b_neg (params/grumpkin.h: b = -17) */
} curve_t;
static const curve_t BN254 = {&BN254_FQ, &BN254_FR, 3, 0};
static const curve_t BLS12_381 = {&BLS_FQ, &BLS_FR, 4, 0};
static const curve_t BLS12_377 = {&BLS377_FQ, &BLS377_FR, 1, 0};
/* Grumpkin: base field = BN254's scalar field, scalar field = BN254's base field (fields/snark_fields/grumpkin_{base,scalar}.h) */
static const curve_t GRUMPKIN = {&BN254_FR, &BN254_FQ, 17, 1};

typedef struct {
  fe_t x, y, z;
} proj_t; /* Projective{x,y,z}, identity (0,1,0): curves/projective.h:26 */
typedef struct {
  fe_t x, y;
} aff_t; /* Affine{x,y}, identity (0,0): curves/affine.h:16,28 */

static void proj_zero(const curve_t* C, proj_t* r)
{
  memset(r, 0, sizeof *r);
  fe_set_u32(C->fq, &r->y, 1);
}
/* mul_weierstrass_b<Gen, true> (fields/field.h:23-56): 3*b*t by a small-constant multiply */
static void mul_b3(const curve_t* C, fe_t* r, const fe_t* t)
{
  fe_t k;
  fe_set_u32(C->fq, &k, 3 * C->b);
  if (C->b_neg) { /* field.h mul_weierstrass_b with is_b_neg: negate */
    fe_t z;
    memset(&z, 0, sizeof z);
    fe_sub(C->fq, &k, &z, &k);
  }
  fe_mul(C->fq, r, t, &k);
}
/* Projective + Projective, complete formula (projective.h:101-143; Renes-Costello-Batina Alg. 7, a = 0) */
static void proj_add(const curve_t* C, proj_t* r, const proj_t* p, const proj_t* q)
{
  const field_t* F = C->fq;
  fe_t t0, t1, t2, t3, t4, t5, a, b, x3, y3, z3;
  fe_mul(F, &t0, &p->x, &q->x);
  fe_mul(F, &t1, &p->y, &q->y);
  fe_mul(F, &t2, &p->z, &q->z);
  fe_add(F, &a, &p->x, &p->y);
  fe_add(F, &b, &q->x, &q->y);
  fe_mul(F, &t3, &a, &b);
  fe_add(F, &a, &t0, &t1);
  fe_sub(F, &t3, &t3, &a);
  fe_add(F, &a, &p->y, &p->z);
  fe_add(F, &b, &q->y, &q->z);
  fe_mul(F, &t4, &a, &b);
  fe_add(F, &a, &t1, &t2);
  fe_sub(F, &t4, &t4, &a);
  fe_add(F, &a, &p->x, &p->z);
  fe_add(F, &b, &q->x, &q->z);
  fe_mul(F, &t5, &a, &b);
  fe_add(F, &a, &t0, &t2);
  fe_sub(F, &t5, &t5, &a);
  fe_add(F, &a, &t0, &t0);
  fe_add(F, &t0, &a, &t0); /* 3*X1X2 */
  mul_b3(C, &t2, &t2);
  fe_add(F, &z3, &t1, &t2);
  fe_sub(F, &t1, &t1, &t2);
  mul_b3(C, &y3, &t5);
  fe_mul(F, &x3, &t4, &y3);
  fe_mul(F, &t2, &t3, &t1);
  fe_sub(F, &x3, &t2, &x3);
  fe_mul(F, &y3, &y3, &t0);
  fe_mul(F, &t1, &t1, &z3);
  fe_add(F, &y3, &t1, &y3);
  fe_mul(F, &t0, &t0, &t3);
  fe_mul(F, &z3, &z3, &t4);
  fe_add(F, &z3, &z3, &t0);
  r->x = x3;
  r->y = y3;
  r->z = z3;
}
/* Projective + Affine (projective.h:147-188): the mixed form of the same complete formula;
 * equivalent to from_affine (projective.h:28-31, zero affine -> zero projective) then add. */
static void proj_from_affine(const curve_t* C, proj_t* r, const aff_t* a)
{
  if (fe_is_zero(C->fq, &a->x) && fe_is_zero(C->fq, &a->y)) {
    proj_zero(C, r);
    return;
  }
  r->x = a->x;
  r->y = a->y;
  fe_set_u32(C->fq, &r->z, 1);
}
static void proj_add_affine(const curve_t* C, proj_t* r, const proj_t* p, const aff_t* a)
{
  proj_t q;
  proj_from_affine(C, &q, a);
  proj_add(C, r, p, &q);
}
static void proj_dbl(const curve_t* C, proj_t* r, const proj_t* p) { proj_add(C, r, p, p); } /* projective.h:73-99 */
static void proj_neg(const curve_t* C, proj_t* r, const proj_t* p)
{
  *r = *p;
  fe_neg(C->fq, &r->y, &p->y);
}
/* Projective::to_affine (projective.h:55-59): x/z, y/z with inverse(0) = 0 => identity -> (0,0) */
static void proj_to_affine(const curve_t* C, aff_t* r, const proj_t* p)
{
  fe_t zi;
  fe_inv(C->fq, &zi, &p->z);
  fe_mul(C->fq, &r->x, &p->x, &zi);
  fe_mul(C->fq, &r->y, &p->y, &zi);
}

static const curve_t* curve_by_id(int id) { return id == 0 ? &BN254 : id == 1 ? &BLS12_381 : id == 2 ? &BLS12_377 : id == 3 ? &GRUMPKIN : 0; }

/* get_scalar_digit (modular_arithmetic.h:280-290): c bits starting at bit digit_num*c */
static uint32_t scalar_digit(const uint32_t* w, int nwords, int digit, int c)
{
  const int bit = digit * c;
  const int word = bit >> 5;
  /* c <= 24 and the in-word shift is < 32, so a 64-bit window suffices */
  uint64_t lo = (word < nwords ? w[word] : 0) | ((uint64_t)(word + 1 < nwords ? w[word + 1] : 0) << 32);
  return (uint32_t)((lo >> (bit & 31)) & ((1u << c) - 1));
}

/*
 * Bucket-method MSM, single worker: cpu_msm.hpp
 *   calc_optimal_parameters :199-223  (nof_bms, bm_size = 2^(c-1); c is supplied by the caller here --
 *                                      the reference picks it with a decision tree, :103-159)
 *   worker_run_phase1       :259-314  (negate scalar+point when the top bit is set :276-277, skip zero
 *                                      base :282, signed digits with carry :289-295)
 *   phase2 / phase3         :317-417  (running "line"/"triangle" sums per bucket module, then Horner
 *                                      with c doublings between modules)
 * scalars: n x 8 u32 canonical; bases: n affine; out: 3*L u32 projective (canonical).
 * Returns 0 on success.
 */
int oracle_msm(int curve_id, const uint32_t* scalars, const uint32_t* bases, int n, int c, int bitsize, uint32_t* out)
{
  const curve_t* C = curve_by_id(curve_id);
  if (!C || c < 1 || c > 24 || n < 0) return -1;
  const field_t* FQ = C->fq;
  const field_t* FR = C->fr;
  const int qw = 2 * FQ->nl, sw = 2 * FR->nl;
  int nbits_full = 0; /* scalar_t::NBITS */
  for (int i = 64 * FR->nl - 1; i >= 0; i--)
    if ((FR->p[i >> 6] >> (i & 63)) & 1) {
      nbits_full = i + 1;
      break;
    }
  const int scalar_size = bitsize ? bitsize : nbits_full;
  const int chopped = scalar_size != nbits_full;
  const int size_with_carry = chopped ? scalar_size + 1 : scalar_size; /* :202-211 */
  const int nof_bms = (size_with_carry - 1) / c + 1;
  const uint32_t bm_size = 1u << (c - 1);
  const size_t nbuckets = (size_t)nof_bms * (bm_size + 1);
  proj_t* buckets = (proj_t*)malloc(nbuckets * sizeof(proj_t));
  unsigned char* busy = (unsigned char*)calloc(nbuckets, 1);
  if (!buckets || !busy) return -2;

  for (int i = 0; i < n; i++) {
    fe_t s;
    fe_load(FR, &s, scalars + (size_t)i * sw);
    aff_t base;
    fe_load(FQ, &base.x, bases + (size_t)i * 2 * qw);
    fe_load(FQ, &base.y, bases + (size_t)i * 2 * qw + qw);
    if (fe_is_zero(FQ, &base.x) && fe_is_zero(FQ, &base.y)) continue; /* :282 */
    int negate = 0;
    if (!chopped && ((s.l[(nbits_full - 1) >> 6] >> ((nbits_full - 1) & 63)) & 1)) { /* :276-277 */
      fe_neg(FR, &s, &s);
      negate = 1;
    }
    uint32_t sword[2 * MAXL + 2];
    memset(sword, 0, sizeof sword);
    fe_store(FR, sword, &s);
    uint32_t carry = 0;
    for (int bm = 0; bm < nof_bms; bm++) {
      uint32_t d = scalar_digit(sword, sw, bm, c) + carry;
      int neg = negate;
      if (d > bm_size) { /* :289-295 */
        d = (1u << c) - d;
        carry = 1;
        neg = !neg;
      } else {
        carry = 0;
      }
      if (d == 0) continue;
      aff_t pt = base;
      if (neg) fe_neg(FQ, &pt.y, &base.y);
      const size_t idx = (size_t)bm * (bm_size + 1) + d;
      if (busy[idx]) {
        proj_add_affine(C, &buckets[idx], &buckets[idx], &pt); /* :297-299 */
      } else {
        proj_from_affine(C, &buckets[idx], &pt); /* :300-304 */
        busy[idx] = 1;
      }
    }
  }
  /* phases 2+3: per bucket module sum_k k*B_k by running sums, then Horner over modules */
  proj_t result;
  proj_zero(C, &result);
  for (int bm = nof_bms - 1; bm >= 0; bm--) {
    for (int k = 0; k < c; k++) proj_dbl(C, &result, &result);
    proj_t line, tri;
    proj_zero(C, &line);
    proj_zero(C, &tri);
    for (uint32_t d = bm_size; d >= 1; d--) {
      const size_t idx = (size_t)bm * (bm_size + 1) + d;
      if (busy[idx]) proj_add(C, &line, &line, &buckets[idx]);
      proj_add(C, &tri, &tri, &line);
    }
    proj_add(C, &result, &result, &tri);
  }
  free(buckets);
  free(busy);
  fe_store(FQ, out, &result.x);
  fe_store(FQ, out + qw, &result.y);
  fe_store(FQ, out + 2 * qw, &result.z);
  return 0;
}

/* projective (3*L words) -> affine (2*L words), Projective::to_affine */
int oracle_to_affine(int curve_id, const uint32_t* proj, uint32_t* aff)
{
  const curve_t* C = curve_by_id(curve_id);
  if (!C) return -1;
  const int qw = 2 * C->fq->nl;
  proj_t p;
  aff_t a;
  fe_load(C->fq, &p.x, proj);
  fe_load(C->fq, &p.y, proj + qw);
  fe_load(C->fq, &p.z, proj + 2 * qw);
  proj_to_affine(C, &a, &p);
  fe_store(C->fq, aff, &a.x);
  fe_store(C->fq, aff + qw, &a.y);
  (void)proj_neg;
  return 0;
}

/* ======================================================================================== NTT */
/* 31-bit fields: stark_fields/babybear.h:10,71 (p = 0x78000001, rou = 0x89, two-adicity 27),
 * koalabear.h:10,71 (p = 0x7f000001, rou = 0x6ac49f88, two-adicity 24). Field ops as in ModArith. */
typedef struct {
  uint32_t p, rou;
  int two_adicity;
} sfield_t;
static const sfield_t BABYBEAR = {0x78000001u, 0x89u, 27};
static const sfield_t KOALABEAR = {0x7f000001u, 0x6ac49f88u, 24};
static const sfield_t* sfield_by_id(int id) { return id == 0 ? &BABYBEAR : id == 1 ? &KOALABEAR : 0; }
static uint32_t s_mul(const sfield_t* F, uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % F->p); }
static uint32_t s_add(const sfield_t* F, uint32_t a, uint32_t b)
{
  uint32_t s = a + b;
  return s >= F->p ? s - F->p : s;
}
static uint32_t s_sub(const sfield_t* F, uint32_t a, uint32_t b) { return a >= b ? a - b : a + F->p - b; }
static uint32_t s_pow(const sfield_t* F, uint32_t a, uint64_t e)
{
  uint32_t r = 1;
  while (e) {
    if (e & 1) r = s_mul(F, r, a);
    a = s_mul(F, a, a);
    e >>= 1;
  }
  return r;
}
static uint32_t bitrev32(uint32_t x, int bits)
{
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}
/* ModArith::omega (modular_arithmetic.h:61-73) */
uint32_t oracle_omega(int field_id, int logn)
{
  const sfield_t* F = sfield_by_id(field_id);
  if (!F || logn > F->two_adicity) return 0;
  if (logn == 0) return 1;
  uint32_t w = F->rou;
  for (int i = 0; i < F->two_adicity - logn; i++) w = s_mul(F, w, w);
  return w;
}

/*
 * NTT as the CPU backend defines it (radix-2 path): backend/cpu/include/
 *   cpu_ntt_domain.h:65-110   twiddles[i] = root^i for the domain the caller initialised (domain_root of
 *                             order 2^log_max); w_N = twiddles[max/N]
 *   ntt_cpu.h:247-306         copy_and_reorder_if_needed: kRN/kRR read the input bit-reversed;
 *                             columns_batch => element j of transform b at j*batch + b (:250,:274-275)
 *   ntt_cpu.h:317-364, :73, :226   coset_mul: forward multiplies x[j] by g^j BEFORE, inverse multiplies by
 *                             g^-j AFTER
 *   ntt_task.h:1162-1238      hierarchy_0_dit_ntt: bit-reverse then radix-2 DIT, twiddle index j*step
 *                             (forward) or max - j*step (inverse); 1/N folded in for kInverse (:1231-1236)
 *   ntt_cpu.h:228-230         kNR/kRR: output bit-reversed. kNM/kMN are treated as kNN.
 * ordering: 0 NN, 1 NR, 2 RN, 3 RR, 4 NM, 5 MN. Returns 0 on success.
 */
int oracle_ntt(int field_id, const uint32_t* in, int n, uint32_t domain_root, int inverse, int ordering, uint32_t coset_gen, int batch, int columns_batch, int lanes, uint32_t* out)
{
  const sfield_t* F = sfield_by_id(field_id);
  if (!F || n <= 0 || (n & (n - 1))) return -1;
  int logn = 0;
  while ((1 << logn) < n) logn++;
  /* order of the domain root by repeated squaring (cpu_ntt_domain.h:78-94) */
  int log_max = 0;
  for (uint32_t x = domain_root; x != 1; x = s_mul(F, x, x)) {
    if (++log_max > F->two_adicity) return -2;
  }
  if (logn > log_max) return -3;
  uint32_t wn = domain_root; /* w_N = root^(max/N) */
  for (int i = 0; i < log_max - logn; i++) wn = s_mul(F, wn, wn);
  if (inverse) wn = s_pow(F, wn, (uint64_t)F->p - 2);
  uint32_t* tw = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n / 2 + 1));
  uint32_t* buf = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  if (!tw || !buf) return -4;
  tw[0] = 1;
  for (int i = 1; i < n / 2; i++) tw[i] = s_mul(F, tw[i - 1], wn);
  const int in_rev = (ordering == 2 || ordering == 3), out_rev = (ordering == 1 || ordering == 3);
  const uint32_t ninv = s_pow(F, (uint32_t)(n % F->p), (uint64_t)F->p - 2);
  const uint32_t ginv = s_pow(F, coset_gen, (uint64_t)F->p - 2);
  const int nb = batch * lanes;
  for (int bp = 0; bp < nb; bp++) {
    const int b = bp / lanes, lane = bp % lanes;
    const size_t bs = columns_batch ? (size_t)lanes : (size_t)n * lanes;
    const size_t es = columns_batch ? (size_t)batch * lanes : (size_t)lanes;
    const size_t off = (size_t)b * bs + lane;
    /* load logical x[j] (input reorder), coset pre-multiplication, then bit-reverse for the DIT */
    uint32_t g = 1;
    for (int j = 0; j < n; j++) {
      const uint32_t src = in_rev ? bitrev32((uint32_t)j, logn) : (uint32_t)j;
      uint32_t v = in[off + (size_t)src * es];
      if (!inverse && coset_gen != 1) {
        v = s_mul(F, v, g);
        g = s_mul(F, g, coset_gen);
      }
      buf[bitrev32((uint32_t)j, logn)] = v;
    }
    for (int half = 1; half < n; half <<= 1) {
      const int step = n / (2 * half);
      for (int i = 0; i < n; i += 2 * half)
        for (int j = 0; j < half; j++) {
          const uint32_t u = buf[i + j], v = s_mul(F, buf[i + j + half], tw[j * step]);
          buf[i + j] = s_add(F, u, v);
          buf[i + j + half] = s_sub(F, u, v);
        }
    }
    g = 1;
    for (int k = 0; k < n; k++) {
      uint32_t v = buf[k];
      if (inverse) {
        v = s_mul(F, v, ninv);
        if (coset_gen != 1) {
          v = s_mul(F, v, g);
          g = s_mul(F, g, ginv);
        }
      }
      const uint32_t dst = out_rev ? bitrev32((uint32_t)k, logn) : (uint32_t)k;
      out[off + (size_t)dst * es] = v;
    }
  }
  free(tw);
  free(buf);
  return 0;
}
