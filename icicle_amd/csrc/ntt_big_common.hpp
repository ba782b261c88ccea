// Shared by ntt_big.hip (scalar-field NTT) and ecntt.hip (NTT over curve points, same twiddle domain):
// field helpers on top of bigfield.hpp, the per-device twiddle domain of a 256-bit scalar field, and
// the twiddle / coset-power generators.
#pragma once
#include "common.h"
#include "bigfield.hpp"
#include <cstring>
#include <map>
#include <mutex>

namespace icicle_hip {

  template <class PR>
  struct BigNtt {
    using F = FieldOps<PR>;
    using fe = typename F::fe;
    static constexpr int W = F::N32; // words per element in memory

    static HD fe load_words(const uint32_t* w) { return F::unpack(w); }
    static HD fe pow_u64(fe base, uint64_t e)
    { // base Montgomery, result Montgomery (canonical limbs not required)
      fe r = F::one();
      bool started = false;
      for (int bit = 63; bit >= 0; bit--) {
        if (started) r = F::sqr(r);
        if ((e >> bit) & 1) {
          r = started ? F::mul(r, base) : base;
          started = true;
        }
      }
      return r;
    }
    static HD fe pow_words(fe base, const uint32_t* e, int nwords)
    {
      fe r = F::one();
      bool started = false;
      for (int bit = nwords * 32 - 1; bit >= 0; bit--) {
        if (started) r = F::sqr(r);
        if ((e[bit >> 5] >> (bit & 31)) & 1) {
          r = started ? F::mul(r, base) : base;
          started = true;
        }
      }
      return r;
    }
    static HD void store_packed(uint32_t* w, const fe& a) { F::pack(w, F::reduce(a)); }
  };

  struct BigDomain {
    uint32_t* tw = nullptr; // tw[i*W .. i*W+W-1] = packed Montgomery w_max^i, i < max_size (W = words per element: 8, or 2 for goldilocks)
    int log_max = -1;
    uint32_t root[8] = {0}; // canonical w_max
    int owner = -1;         // >= 0: brought up by a multi-device call of device `owner` (released with its domain)
  };
  template <class PR>
  struct BigDomainStore {
    static std::mutex& mtx()
    {
      static std::mutex m;
      return m;
    }
    static std::map<int, BigDomain>& map()
    {
      static std::map<int, BigDomain> m;
      return m;
    }
  };

  struct BigWords {
    uint32_t w[8];
  };

  // tw[i] = root^i. Thread t fills 64 consecutive entries from one pow().
  template <class PR>
  __global__ __launch_bounds__(256) void k_big_gen_twiddles(uint32_t* __restrict__ tw, BigWords root_mont, size_t n)
  {
    using B = BigNtt<PR>;
    using F = typename B::F;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t i0 = t * 64;
    if (i0 >= n) return;
    const typename B::fe r = F::unpack(root_mont.w);
    typename B::fe x = B::pow_u64(r, (uint64_t)i0);
    for (size_t i = i0; i < i0 + 64 && i < n; i++) {
      B::store_packed(tw + i * B::W, x);
      x = F::mul(x, r);
    }
  }

  // pw[i] = g^i (packed Montgomery), i < n
  template <class PR>
  __global__ __launch_bounds__(256) void k_big_coset_powers(uint32_t* __restrict__ pw, BigWords g_mont, uint64_t n)
  {
    using B = BigNtt<PR>;
    using F = typename B::F;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t i0 = t * 16;
    if (i0 >= n) return;
    const typename B::fe g = F::unpack(g_mont.w);
    typename B::fe x = B::pow_u64(g, i0);
    for (uint64_t i = i0; i < i0 + 16 && i < n; i++) {
      B::store_packed(pw + i * B::W, x);
      x = F::mul(x, g);
    }
  }

  __device__ __forceinline__ void load8(uint32_t* dst, const uint32_t* __restrict__ src)
  {
    const uint4 a = ((const uint4*)src)[0], b = ((const uint4*)src)[1];
    dst[0] = a.x, dst[1] = a.y, dst[2] = a.z, dst[3] = a.w;
    dst[4] = b.x, dst[5] = b.y, dst[6] = b.z, dst[7] = b.w;
  }
  __device__ __forceinline__ void store8(uint32_t* __restrict__ dst, const uint32_t* src)
  {
    ((uint4*)dst)[0] = make_uint4(src[0], src[1], src[2], src[3]);
    ((uint4*)dst)[1] = make_uint4(src[4], src[5], src[6], src[7]);
  }

  // W words of one element (8: the 256-bit fields; 2: goldilocks)
  template <int W>
  __device__ __forceinline__ void loadw(uint32_t* dst, const uint32_t* __restrict__ src)
  {
    if constexpr (W == 8) {
      load8(dst, src);
    } else {
      static_assert(W == 2, "element width");
      const uint2 a = ((const uint2*)src)[0];
      dst[0] = a.x, dst[1] = a.y;
    }
  }
  template <int W>
  __device__ __forceinline__ void storew(uint32_t* __restrict__ dst, const uint32_t* src)
  {
    if constexpr (W == 8) {
      store8(dst, src);
    } else {
      static_assert(W == 2, "element width");
      ((uint2*)dst)[0] = make_uint2(src[0], src[1]);
    }
  }

  // ---- host ------------------------------------------------------------------------------------
  // (w holds PR::NL32 words of the caller's)
  template <class PR>
  static bool words_lt_p(const uint32_t* w)
  {
    for (int i = PR::NL32 - 1; i >= 0; i--) {
      if (w[i] < PR::P32[i]) return true;
      if (w[i] > PR::P32[i]) return false;
    }
    return false;
  }
  static inline bool words_is_zero(const uint32_t* w, int nw = 8)
  {
    uint32_t o = 0;
    for (int i = 0; i < nw; i++)
      o |= w[i];
    return o == 0;
  }
  static inline bool words_is_one(const uint32_t* w, int nw = 8)
  {
    uint32_t o = w[0] ^ 1u;
    for (int i = 1; i < nw; i++)
      o |= w[i];
    return o == 0;
  }
  template <class PR>
  static BigWords mont_words(const typename FieldOps<PR>::fe& x)
  {
    BigWords r{};
    BigNtt<PR>::store_packed(r.w, x);
    return r;
  }
  template <class PR>
  static typename FieldOps<PR>::fe host_inverse(const typename FieldOps<PR>::fe& x)
  { // x^(p-2)
    uint32_t e[8];
    uint64_t borrow = 2;
    for (int i = 0; i < 8; i++) {
      const uint64_t v = (uint64_t)PR::P32[i];
      const uint64_t d = v - borrow;
      e[i] = (uint32_t)d;
      borrow = (v < borrow) ? 1 : 0;
    }
    return BigNtt<PR>::pow_words(x, e, 8);
  }

  // (1/2)^logn in Montgomery form; 1/2 = (p+1)/2
  template <class PR>
  static typename FieldOps<PR>::fe host_ninv(int logn)
  {
    uint32_t h[8];
    uint64_t c = 1;
    for (int i = 0; i < 8; i++) {
      const uint64_t v = (uint64_t)PR::P32[i] + c;
      h[i] = (uint32_t)v;
      c = v >> 32;
    }
    for (int i = 0; i < 8; i++)
      h[i] = (h[i] >> 1) | (i < 7 ? (h[i + 1] << 31) : ((uint32_t)c << 31));
    return BigNtt<PR>::pow_u64(FieldOps<PR>::from_canonical(h), (uint64_t)logn);
  }

} // namespace icicle_hip
