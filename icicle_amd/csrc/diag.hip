// Diagnostics exported next to the product entry points so that bench.py can measure the roofs of the dominant MSM
// kernel IN THE SAME RUN as the kernel itself (no hard-coded ceilings):
//   icicle_hip_ubench_mixed_add  XYZZ mixed adds per second with every operand in registers -- the integer-ALU roof
//                                of k_accumulate (same ec.hpp code, same launch bounds, no memory traffic)
//   icicle_hip_ubench_gather     random 64-byte gathers per second (4 x 16 B loads per lane, the access pattern of
//                                k_accumulate's base fetch) over a region of the given size; also the known-byte-count
//                                kernel the FETCH_SIZE counter is calibrated on (tools/pmc_traffic.sh)
//   icicle_hip_ubench_ntt_pass   the arithmetic of one 8-stage NTT pass per second with every operand in registers: per 16 elements two
//                                radix-16 register rounds (ntt_fast.hpp ntt_stages, the code of k_ntt_fast) and the 16 products with the
//                                factor behind the pass -- the VALU-issue roof of a transform, no memory, no LDS
// None touches user data. tools/ubench/msm_ubench.hip holds the wider design-decision sweeps.
#include "common.h"
#include "ec.hpp"
#include "ec_dbl_quad.hpp"
#include "ntt_fast.hpp"

namespace icicle_hip {

  __device__ __forceinline__ uint32_t diag_mix(uint32_t x)
  {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
  }

  template <class C>
  __global__ __launch_bounds__(128, (EC<C>::F::N <= 9 ? 3 : 2)) void k_diag_madd(uint32_t* __restrict__ out, int iters, uint32_t seed)
  {
    using E = EC<C>;
    using F = typename E::F;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    // The loop is k_accumulate's own (msm_impl.hpp): an empty accumulator, a fresh operand every step -- built from two live
    // words where the real kernel gathers it from HBM --, load_plain + cneg + madd, to_proj at the end. (Rounds 1-5 kept a
    // seeded accumulator and the operand live across the addition: 18 registers more than the real kernel, 144 B of scratch at
    // the same launch bounds -- a roof measured on a spilling kernel reads low.)
    constexpr int PW = 2 * E::N32;
    typename E::XYZZ acc;
    bool empty = true;
    uint32_t sx = diag_mix(t + seed), sy = diag_mix(t * 3 + seed + 13);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
      sx += 0x1234567u, sy ^= (uint32_t)(it + 1) * 0x9e3779b9u;
      uint32_t w[PW];
#pragma unroll
      for (int k = 0; k < E::N32; k++) {
        w[k] = sx + (uint32_t)k * 0x3c6ef37u;
        w[E::N32 + k] = sy ^ ((uint32_t)k * 0x51ed27fu);
      }
      w[E::N32 - 1] &= 0x00ffffffu, w[PW - 1] &= 0x00ffffffu; // below p
      E::madd(acc, empty, E::cneg(E::load_plain(w), (it & 1) != 0));
    }
    const typename E::Proj pr = E::to_proj(acc, empty);
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++)
      r ^= pr.x.l[i] ^ pr.y.l[i] ^ pr.z.l[i];
    out[t] = r;
  }

  __global__ __launch_bounds__(256) void k_diag_gather(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t region_pts, uint32_t per_thread, uint32_t seed)
  {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    uint32_t s = diag_mix(t * 2654435761u + seed);
    for (uint32_t i = 0; i < per_thread; i++) {
      s = diag_mix(s + 0x9e3779b9u);
      const uint4* p = in + (size_t)(((uint64_t)s * region_pts) >> 32) * 4;
      const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
      acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
      acc.y += a.y + b.x;
      acc.z ^= c.x;
      acc.w += d.x;
    }
    out[t] = acc;
  }

  // One "pass unit" = 16 elements through an 8-stage pass: radix-16 round, radix-16 round, product with the inter-pass factor. Same
  // block shape and register budget as the two-round k_ntt_fast instantiations (512 threads, 4 waves per SIMD).
  template <class PR>
  __global__ __launch_bounds__(512, 4) void k_diag_ntt_pass(uint32_t* __restrict__ out, int iters, uint32_t seed)
  {
    using S = SmallField<PR>;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[16], w0[15], w1[15], wip[16];
    uint32_t s = diag_mix(t + seed);
#pragma unroll
    for (int i = 0; i < 16; i++) {
      s = diag_mix(s + i);
      x[i] = s % PR::P;
      s = diag_mix(s + 77);
      wip[i] = s % PR::P;
    }
#pragma unroll
    for (int i = 0; i < 15; i++) {
      s = diag_mix(s + 3);
      w0[i] = s % PR::P;
      s = diag_mix(s + 5);
      w1[i] = s % PR::P;
    }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
      ntt_stages<S, 4, false, true>(x, w0);
      ntt_stages<S, 4, false, true>(x, w1);
#pragma unroll
      for (int m = 0; m < 16; m++)
        x[m] = S::mul(x[m], wip[m]);
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
      r ^= x[i];
    out[t] = r;
  }

  template <class PR>
  static icicle_error_t ntt_pass_bench(double* rate)
  {
    ICICLE_TRY(bind_current_device());
    const int blocks = 256 * 16, iters = 512;
    TempBuf o;
    HIP_TRY(o.alloc((size_t)blocks * 512 * 4, nullptr), ICICLE_ALLOCATION_FAILED);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipEventCreate(&e1), ICICLE_INVALID_ARGUMENT);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, nullptr);
      k_diag_ntt_pass<PR><<<blocks, 512>>>(o.as<uint32_t>(), iters, 11 + rep);
      (void)hipEventRecord(e1, nullptr);
      HIP_TRY(hipEventSynchronize(e1), ICICLE_SYNCHRONIZATION_FAILED);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *rate = (double)blocks * 512 * iters / (best * 1e-3);
    return ICICLE_SUCCESS;
  }

  template <class C>
  static icicle_error_t madd_bench(double* rate)
  {
    ICICLE_TRY(bind_current_device());
    const int blocks = 256 * 24, iters = 256;
    TempBuf o;
    HIP_TRY(o.alloc((size_t)blocks * 128 * 4, nullptr), ICICLE_ALLOCATION_FAILED);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipEventCreate(&e1), ICICLE_INVALID_ARGUMENT);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, nullptr);
      k_diag_madd<C><<<blocks, 128>>>(o.as<uint32_t>(), iters, 7 + rep);
      (void)hipEventRecord(e1, nullptr);
      HIP_TRY(hipEventSynchronize(e1), ICICLE_SYNCHRONIZATION_FAILED);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *rate = (double)blocks * 128 * iters / (best * 1e-3);
    return ICICLE_SUCCESS;
  }

  // ---- self-test of the in-place asm products: aliased and constant operands ----------------------------
  // mul_inplace / mul_add_inplace_c overwrite their read-write operand while the other operands are still being read;
  // the read-write operands are early-clobber, so an input that is the SAME value (a squaring written as mul_inplace(a, a))
  // or a compile-time constant must still give the value of the out-of-place product. Every thread checks a different
  // pseudo-random operand set; mismatches are counted.
  template <class C>
  __global__ __launch_bounds__(64) void k_selftest_inplace(uint32_t* __restrict__ mismatches, uint32_t seed)
  {
    using F = typename EC<C>::F;
    using fe = typename F::fe;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    auto seed_fe = [&](uint32_t s) {
      fe r;
#pragma unroll
      for (int i = 0; i < F::N; i++) {
        s = diag_mix(s + i);
        r.l[i] = s & RB_MASK;
      }
      r.l[F::N - 1] &= 0xfffff; // well below p
      BF_SET_BOUND(r, 1);
      return r;
    };
    auto same = [&](const fe& x, const fe& y) {
      uint32_t d = 0;
#pragma unroll
      for (int i = 0; i < F::N; i++)
        d |= x.l[i] ^ y.l[i];
      return d == 0;
    };
    const fe a = seed_fe(t * 7 + seed), b = seed_fe(t * 11 + seed + 1), c = seed_fe(t * 13 + seed + 2), d = seed_fe(t * 17 + seed + 3);
    uint32_t bad = 0;
    { // plain: in place == out of place
      fe x = a;
      F::mul_inplace(x, b);
      bad += !same(x, F::mul(a, b));
    }
    { // aliased: a <- a * a
      fe x = a;
      F::mul_inplace(x, x);
      bad += !same(x, F::mul(a, a));
    }
    { // constant operands (R^2 and the Montgomery one share limbs with each other and with zero limbs)
      fe x = a;
      F::mul_inplace(x, F::r2());
      bad += !same(x, F::mul(a, F::r2()));
      fe y = F::one();
      F::mul_inplace(y, F::one());
      bad += !same(y, F::mul(F::one(), F::one()));
    }
    { // c <- a*b + c*d, plain and with every aliasing of the read-write operand
      fe x = c;
      F::mul_add_inplace_c(x, a, b, d);
      bad += !same(x, F::mul_add(a, b, c, d));
      fe y = c;
      F::mul_add_inplace_c(y, y, b, d); // a aliases c
      bad += !same(y, F::mul_add(c, b, c, d));
      fe z = c;
      F::mul_add_inplace_c(z, a, z, z); // b and d alias c
      bad += !same(z, F::mul_add(a, c, c, c));
    }
    if (bad) atomicAdd(mismatches, bad);
  }

  template <class C>
  static icicle_error_t selftest_inplace_run(int* mismatches)
  {
    ICICLE_TRY(bind_current_device());
    TempBuf cnt;
    HIP_TRY(cnt.alloc(16, nullptr), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(hipMemsetAsync(cnt.ptr(), 0, 16, nullptr), ICICLE_COPY_FAILED);
    k_selftest_inplace<C><<<64, 64>>>(cnt.as<uint32_t>(), 12345u);
    LAUNCH_CHECK("k_selftest_inplace", nullptr);
    uint32_t h = 0;
    HIP_TRY(hipMemcpy(&h, cnt.ptr(), 4, hipMemcpyDeviceToHost), ICICLE_COPY_FAILED);
    *mismatches = (int)h;
    return ICICLE_SUCCESS;
  }

  // ---- self-test of the quad-cooperative group operations (ec.hpp add_quad, ec_dbl_quad.hpp EcQuadAdd / EcDblSmallB) -----------
  // The ECNTT's butterflies and the MSM's window combine spread the complete addition and doubling over the four lanes of a DPP
  // quad. Here every quad takes a pair (P, Q) = (a G, +- b G), a, b in 0..6 (0 = the identity) -- so P = Q, P = -Q, O + P, P + O and
  // O + O all occur -- and compares each quad form with the one-lane complete formulas (add_body / dbl_body, themselves checked against
  // Python integers on the host) AS GROUP ELEMENTS: cross-multiplied coordinates, both or neither at infinity.
  template <class C>
  __global__ __launch_bounds__(64) void k_selftest_quad_ops(uint32_t* __restrict__ mismatches)
  {
    using E = EC<C>;
    using F = typename E::F;
    using Proj = typename E::Proj;
    const uint32_t role = threadIdx.x & 3u;
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const uint32_t a = t % 7u, b = (t / 7u) % 7u, neg = (t / 49u) & 1u;
    const Proj g = E::to_proj(E::generator());
    const Proj p = E::mul_small(g, a);
    Proj q = E::mul_small(g, b);
    if (neg) q.y = F::template neg<4>(q.y);
    auto same_point = [&](const Proj& u, const Proj& v) {
      const bool zu = F::is_zero(u.z), zv = F::is_zero(v.z);
      if (zu || zv) return zu == zv && !F::is_zero(u.y) && !F::is_zero(v.y); // (0 : y != 0 : 0), never (0, 0, 0)
      return F::eq(F::mul(u.x, v.z), F::mul(v.x, u.z)) && F::eq(F::mul(u.y, v.z), F::mul(v.y, u.z));
    };
    uint32_t bad = 0;
    const Proj s = E::add(p, q);
    bad += !same_point(E::add_quad(p, q, role), s);
    bad += !same_point(EcQuadAdd<C>::add(p, q, role), s);
    Proj d5 = p, h5 = p;
    for (int i = 0; i < 5; i++) {
      d5 = EcDblSmallB<C>::dbl_quad(d5, role);
      h5 = E::dbl(h5);
    }
    bad += !same_point(EcDblSmallB<C>::dbl_quad(q, role), E::dbl(q));
    bad += !same_point(d5, h5);
    bad += !same_point(d5, E::mul_small(p, 32));
    // a doubling and an addition fed by quad results (the bounds the chain runs with)
    bad += !same_point(EcQuadAdd<C>::add(EcDblSmallB<C>::dbl_quad(s, role), EcQuadAdd<C>::add(p, q, role), role), E::add(E::dbl(s), s));
    if (bad && role == 0) atomicAdd(mismatches, bad);
    if (role != 0 && bad) atomicAdd(mismatches + 1, bad); // (every lane of a quad must hold the result: lanes 1..3 counted apart)
  }

  template <class C>
  static icicle_error_t selftest_quad_ops_run(int* mismatches)
  {
    ICICLE_TRY(bind_current_device());
    TempBuf cnt;
    HIP_TRY(cnt.alloc(16, nullptr), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(hipMemsetAsync(cnt.ptr(), 0, 16, nullptr), ICICLE_COPY_FAILED);
    k_selftest_quad_ops<C><<<(98 * 4 + 63) / 64, 64>>>(cnt.as<uint32_t>()); // 7 x 7 x 2 pairs (the last block wraps around: t % 7 ...)
    LAUNCH_CHECK("k_selftest_quad_ops", nullptr);
    uint32_t h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, cnt.ptr(), 8, hipMemcpyDeviceToHost), ICICLE_COPY_FAILED);
    *mismatches = (int)(h[0] + h[1]);
    return ICICLE_SUCCESS;
  }

} // namespace icicle_hip

using namespace icicle_hip;

extern "C" icicle_error_t icicle_hip_ubench_mixed_add(int curve, double* adds_per_second)
{
  if (!adds_per_second) return ICICLE_INVALID_POINTER;
  try {
    if (curve == 0) return madd_bench<bn254_g1>(adds_per_second);
    if (curve == 1) return madd_bench<bls12_381_g1>(adds_per_second);
  } catch (...) {
  }
  return ICICLE_INVALID_ARGUMENT;
}

// field 0 = BabyBear, 1 = KoalaBear; *pass_units_per_second: one unit = 16 elements through one 8-stage pass (2 radix-16 register
// rounds + 16 inter-pass products). A 2^24-point transform of 3 passes is 3 * 2^20 units per row.
extern "C" icicle_error_t icicle_hip_ubench_ntt_pass(int field, double* pass_units_per_second)
{
  if (!pass_units_per_second) return ICICLE_INVALID_POINTER;
  try {
    if (field == 0) return ntt_pass_bench<babybear_params>(pass_units_per_second);
    if (field == 1) return ntt_pass_bench<koalabear_params>(pass_units_per_second);
  } catch (...) {
  }
  return ICICLE_INVALID_ARGUMENT;
}

extern "C" icicle_error_t icicle_hip_ubench_gather(uint64_t region_bytes, uint64_t gathers, double* gathers_per_second)
{
  if (!gathers_per_second || region_bytes < 64 || gathers == 0) return ICICLE_INVALID_ARGUMENT;
  try {
    ICICLE_TRY(bind_current_device());
    const int blocks = 256 * 8;
    const uint32_t per_thread = (uint32_t)std::max<uint64_t>(1, gathers / ((uint64_t)blocks * 256));
    TempBuf region, o;
    HIP_TRY(region.alloc(region_bytes, nullptr), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(o.alloc((size_t)blocks * 256 * 16, nullptr), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(hipMemsetAsync(region.ptr(), 1, region_bytes, nullptr), ICICLE_COPY_FAILED);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipEventCreate(&e1), ICICLE_INVALID_ARGUMENT);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, nullptr);
      k_diag_gather<<<blocks, 256>>>(region.as<uint4>(), o.as<uint4>(), (uint32_t)(region_bytes / 64), per_thread, 3 + rep);
      (void)hipEventRecord(e1, nullptr);
      HIP_TRY(hipEventSynchronize(e1), ICICLE_SYNCHRONIZATION_FAILED);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gathers_per_second = (double)blocks * 256 * per_thread / (best * 1e-3);
    return ICICLE_SUCCESS;
  } catch (...) {
    return ICICLE_INVALID_ARGUMENT;
  }
}

// In-place field products against their out-of-place forms with aliased and constant operands (curve 0 = bn254's Fq,
// 1 = bls12_381's Fq): *mismatches must come back 0.
extern "C" icicle_error_t icicle_hip_selftest_inplace_products(int curve, int* mismatches)
{
  if (!mismatches) return ICICLE_INVALID_POINTER;
  try {
    if (curve == 0) return selftest_inplace_run<bn254_g1>(mismatches);
    if (curve == 1) return selftest_inplace_run<bls12_381_g1>(mismatches);
  } catch (...) {
  }
  return ICICLE_INVALID_ARGUMENT;
}

// Quad-cooperative complete addition / doubling (ec.hpp add_quad, ec_dbl_quad.hpp) against the one-lane formulas on every pair
// (a G, +- b G), a, b = 0..6, as group elements (curve 0 = bn254, 1 = bls12_381, 2 = bls12_377): *mismatches must come back 0.
extern "C" icicle_error_t icicle_hip_selftest_quad_group_ops(int curve, int* mismatches)
{
  if (!mismatches) return ICICLE_INVALID_POINTER;
  try {
    if (curve == 0) return selftest_quad_ops_run<bn254_g1>(mismatches);
    if (curve == 1) return selftest_quad_ops_run<bls12_381_g1>(mismatches);
    if (curve == 2) return selftest_quad_ops_run<bls12_377_g1>(mismatches);
  } catch (...) {
  }
  return ICICLE_INVALID_ARGUMENT;
}
