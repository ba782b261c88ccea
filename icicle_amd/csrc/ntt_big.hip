// NTT for gfx950 over the 256-bit scalar fields of the curves (BN254 Fr, BLS12-381 Fr).
//
// Reference semantics are those of ntt.hip (icicle/backend/cpu/include/ntt_cpu.h:70-364,
// cpu_ntt_domain.h:65-110,614-654, src/ntt.cpp:11-84); only the element type differs: scalar_t is
// 8 little-endian u32 words, NTTConfig<scalar_t> is 64 bytes (coset_gen is a whole element).
//
// Same pass decomposition as the 31-bit NTT (ntt_plan.h): <= 3 passes of 2^s-point sub-transforms
// on [L x T] tiles held in LDS, inter-pass twiddle on the way out, natural-order scatter in the
// last pass. What changes with 32-byte elements:
//   * a run of T = 4 elements is already a 128 B HBM transaction, so tiles are narrow and tall;
//   * the tile lives in LDS as 9 x 29-bit limbs per element (36 B, odd word stride) so the
//     butterflies use bigfield.hpp's lazy arithmetic without re-packing between stages;
//   * a butterfly is one 9-limb Montgomery product (~230 VALU ops), two stages per LDS round trip: the pass is ALU-bound, not
//     HBM-bound (2^24 elements: ~1.5 GB of traffic per direction vs ~6e10 lane-ops).
// Data stays in the caller's (canonical or Montgomery) form: twiddles are kept in Montgomery form
// and montmul(x, w*R) = x*w. Lazy bounds (units of p): loads 1.2, 3.2 and 7.2 after the two
// product-free stages, +2 per later stage, <= 23.2 after ten stages (limit 64, reduce() accepts
// < 32), back to canonical on every store.
#include "ntt_big_common.hpp"
#include "goldfield.hpp"
#include "ntt_plan.h"
#include "ntt_multi.hpp"
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>

namespace icicle_hip {

  // One pass: grid = (ntiles, batch). Dynamic LDS: L*T elements of 9 limbs.
  // Handles every ordering, cosets, both directions and both batch layouts.
  template <class PR>
  __global__ __launch_bounds__(512) void k_big_ntt_pass(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ tw, const uint32_t* __restrict__ coset_pow, PassDesc pd, NttLaunch nl, BigWords ninv_mont)
  {
    using B = BigNtt<PR>;
    using F = typename B::F;
    using fe = typename B::fe;
    extern __shared__ uint32_t lds_raw[];
    fe* tile = reinterpret_cast<fe*>(lds_raw);
    const uint32_t L = 1u << pd.s, T = pd.T;
    constexpr int W = B::W; // words per element in memory
    const uint32_t bq = nl.row0 + blockIdx.y; // lane-transform index (launched in slices of <= 65535)
    const uint64_t boff = (uint64_t)(bq / nl.lanes) * nl.bs + (bq % nl.lanes); // lanes > 1: components of an extension-field element
    const uint32_t a = blockIdx.x / pd.tiles_per_a, ct = blockIdx.x % pd.tiles_per_a;
    const uint64_t in_base = (uint64_t)a * pd.in_base_a + (uint64_t)ct * pd.in_base_ct;
    const uint64_t max_mask = ((uint64_t)1 << nl.log_max) - 1;
    const uint32_t tot = L * T;
    uint32_t wbuf[8];

    auto tw_load = [&](uint64_t idx) -> fe {
      idx &= max_mask;
      if (nl.inverse) idx = (((uint64_t)1 << nl.log_max) - idx) & max_mask;
      uint32_t w[8];
      loadw<W>(w, tw + idx * W);
      return F::unpack(w);
    };

    // ---- load: logical (k,t) -> LDS row bitrev_s(k)
    for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {
      const uint32_t t = e % T, k = e / T;
      const uint64_t addr = in_base + (uint64_t)k * pd.in_sk + (uint64_t)t * pd.in_st;
      const uint64_t maddr = (nl.in_rev && pd.pidx == 0) ? bitrev64(addr, nl.logn) : addr;
      loadw<W>(wbuf, in + (boff + maddr * nl.es) * W);
      fe v = F::unpack(wbuf);
      if (nl.coset && !nl.inverse && pd.pidx == 0) {
        uint32_t cw[8];
        loadw<W>(cw, coset_pow + addr * W);
        v = F::mul(v, F::unpack(cw));
      }
      tile[(pd.s == 0 ? 0u : (__brev(k) >> (32 - pd.s))) * T + t] = v;
    }
    __syncthreads();

    // ---- s radix-2 DIT stages in LDS, two per round trip; w_L^e = tw[e * (max/L)]
    // A thread takes the four elements i, i+h, i+2h, i+3h (h = 2^q) through stages q and q+1 in registers: half the
    // LDS traffic, index arithmetic and barriers of one stage per trip. Lazy bounds as before: the first two stages
    // skip the trivial-twiddle products (1.2 -> 3.2 -> 7.2), every later stage adds 2.
    const uint32_t lstride_log = nl.log_max - pd.s;
    auto stage_tw = [&](int q, uint32_t pos) -> fe { return tw_load(((uint64_t)pos << (pd.s - 1 - q)) << lstride_log); };
    int q = 0;
    if constexpr (B::W == 2) {
      // 8-byte elements with exact arithmetic (goldilocks): FOUR stages per LDS round trip on 16 elements per thread -- with a
      // product this cheap the pass was bound by its LDS round trips (two stages each: 0.71 ms at 2^24, four times its HBM
      // traffic floor). Element m of the group sits at row i + m h; in stage q + j it is the lower / upper end of a butterfly
      // by bit j of m, and its block position is pos + (m mod 2^j) h.
      for (; q + 3 < pd.s; q += 4) {
        const uint32_t h = 1u << q;
        for (uint32_t id = threadIdx.x; id < tot / 16; id += blockDim.x) {
          const uint32_t t = id % T, bf = id / T;
          const uint32_t pos = bf & (h - 1);
          const uint32_t i = ((bf >> q) << (q + 4)) + pos;
          fe x[16];
#pragma unroll
          for (int m = 0; m < 16; m++)
            x[m] = tile[(i + (uint32_t)m * h) * T + t];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            fe w[8]; // twiddles of stage q + j: one per value of (m mod 2^j)
#pragma unroll
            for (int lo = 0; lo < (1 << j); lo++)
              w[lo] = stage_tw(q + j, pos + (uint32_t)lo * h);
#pragma unroll
            for (int m = 0; m < 16; m++) {
              if (m & (1 << j)) continue;
              const int lo = m & ((1 << j) - 1);
              const fe u = x[m];
              const fe v = (q == 0 && lo == 0) ? x[m + (1 << j)] : F::mul(x[m + (1 << j)], w[lo]); // (q = 0, lo = 0: w^0 = 1)
              x[m] = F::add(u, v);
              x[m + (1 << j)] = F::template sub<2>(u, v);
            }
          }
#pragma unroll
          for (int m = 0; m < 16; m++)
            tile[(i + (uint32_t)m * h) * T + t] = x[m];
        }
        __syncthreads();
      }
    }
    for (; q + 1 < pd.s; q += 2) {
      const uint32_t h = 1u << q;
      for (uint32_t id = threadIdx.x; id < tot / 4; id += blockDim.x) {
        const uint32_t t = id % T, bf = id / T;
        const uint32_t pos = bf & (h - 1);
        const uint32_t i = ((bf >> q) << (q + 2)) + pos;
        fe x0 = tile[i * T + t], x1 = tile[(i + h) * T + t], x2 = tile[(i + 2 * h) * T + t], x3 = tile[(i + 3 * h) * T + t];
        if (q == 0) {
          // stage 0: twiddle 1 everywhere; stage 1: twiddle 1 on (x0, x2), w^(L/4) on (x1, x3)
          fe a0 = F::add(x0, x1), a1 = F::template sub<2>(x0, x1);
          fe a2 = F::add(x2, x3), a3 = F::template sub<2>(x2, x3);
          a3 = F::mul(a3, stage_tw(1, 1));
          x0 = F::add(a0, a2);
          x2 = F::template sub<4>(a0, a2);
          x1 = F::add(a1, a3);
          x3 = F::template sub<4>(a1, a3);
        } else {
          const fe w = stage_tw(q, pos);
          x1 = F::mul(x1, w);
          x3 = F::mul(x3, w);
          fe a0 = F::add(x0, x1), a1 = F::template sub<2>(x0, x1);
          fe a2 = F::add(x2, x3), a3 = F::template sub<2>(x2, x3);
          a2 = F::mul(a2, stage_tw(q + 1, pos));
          a3 = F::mul(a3, stage_tw(q + 1, pos + h));
          x0 = F::add(a0, a2);
          x2 = F::template sub<2>(a0, a2);
          x1 = F::add(a1, a3);
          x3 = F::template sub<2>(a1, a3);
        }
        tile[i * T + t] = x0;
        tile[(i + h) * T + t] = x1;
        tile[(i + 2 * h) * T + t] = x2;
        tile[(i + 3 * h) * T + t] = x3;
      }
      __syncthreads();
    }
    if (q < pd.s) { // odd number of stages: the last one alone
      const uint32_t half = 1u << q;
      for (uint32_t id = threadIdx.x; id < tot / 2; id += blockDim.x) {
        const uint32_t t = id % T, bf = id / T;
        const uint32_t pos = bf & (half - 1);
        const uint32_t i = ((bf >> q) << (q + 1)) + pos;
        const fe u = tile[i * T + t];
        fe v = tile[(i + half) * T + t];
        if (q >= 1 && !(q == 1 && pos == 0)) v = F::mul(v, stage_tw(q, pos));
        tile[i * T + t] = F::add(u, v);
        tile[(i + half) * T + t] = (q == 1) ? F::template sub<4>(u, v) : F::template sub<2>(u, v);
      }
      __syncthreads();
    }

    // ---- store
    // 8-byte elements (goldilocks): a thread's rows k = tid / T + it * (blockDim / T) are an arithmetic progression at a fixed
    // column, so its inter-pass factors are a geometric one -- two table reads and a product per element instead of a gather per
    // element (in pass 1 every gather is a 64-byte sector for 8 bytes; same idea as ntt_fast.hpp). Needs blockDim % T == 0 (T and
    // blockDim are powers of two, T <= 16 <= 64 <= blockDim). The 256-bit fields keep the reads: a product costs them 163 mads.
    fe ip_cur = F::one(), ip_ratio = F::one();
    if constexpr (B::W == 2) {
      if (!pd.is_last) {
        const uint32_t t = threadIdx.x % T, k0 = threadIdx.x / T, kstep = blockDim.x / T;
        const uint64_t jnext = ((uint64_t)ct * T + t) / pd.cprime;
        const uint64_t A = (pd.pidx == 0) ? 0u : (uint64_t)a, Bf = (pd.pidx == 0) ? 1u : (uint64_t)pd.n0;
        ip_cur = tw_load(jnext * (A + Bf * k0) * pd.tw_stride);
        ip_ratio = tw_load(jnext * Bf * kstep * pd.tw_stride);
      }
    }
    for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {
      const uint32_t t = e % T, k = e / T;
      fe v = tile[k * T + t];
      uint64_t oaddr;
      if (!pd.is_last) {
        oaddr = in_base + (uint64_t)k * pd.in_sk + (uint64_t)t * pd.in_st;
        if constexpr (B::W == 2) {
          v = F::mul(v, ip_cur);
          ip_cur = F::mul(ip_cur, ip_ratio);
        } else {
          const uint64_t c = (uint64_t)ct * T + t;
          const uint64_t jnext = c / pd.cprime;
          const uint64_t K = (pd.pidx == 0) ? (uint64_t)k : ((uint64_t)a + (uint64_t)pd.n0 * k);
          v = F::mul(v, tw_load(jnext * K * pd.tw_stride));
        }
      } else {
        uint64_t K0;
        if (pd.pidx <= 1) {
          K0 = (uint64_t)ct * T + t;
        } else {
          const uint64_t k0 = (uint64_t)ct * T + t, k1 = a;
          K0 = k0 + (uint64_t)pd.n0 * k1;
        }
        oaddr = K0 + (uint64_t)k * pd.out_sk;
        if (nl.inverse) {
          v = F::mul(v, F::unpack(ninv_mont.w));
          if (nl.coset) {
            uint32_t cw[8];
            loadw<W>(cw, coset_pow + oaddr * W);
            v = F::mul(v, F::unpack(cw));
          }
        }
        if (nl.out_rev) oaddr = bitrev64(oaddr, nl.logn);
      }
      B::store_packed(wbuf, v);
      storew<W>(out + (boff + oaddr * nl.es) * W, wbuf);
    }
  }

  template <class PR>
  static icicle_error_t big_init_domain_run(const uint32_t* root, const icicle_ntt_init_domain_config_t* cfg)
  {
    using F = FieldOps<PR>;
    if (!root || !cfg) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    const int dev = current_device_id();
    std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
    auto& dom = BigDomainStore<PR>::map()[dev];
    if (dom.tw) return ICICLE_SUCCESS; // already initialised: silent success (cpu_ntt_domain.h:69)
    if (words_is_zero(root, PR::NL32) || !words_lt_p<PR>(root)) return ICICLE_INVALID_ARGUMENT;
    // order of the root by repeated squaring (cpu_ntt_domain.h:78-94)
    const typename F::fe r = F::from_canonical(root);
    typename F::fe x = r;
    int log_max = 0;
    while (!F::eq(x, F::one()) && log_max <= PR::TWO_ADICITY) {
      x = F::sqr(x);
      log_max++;
    }
    if (!F::eq(x, F::one())) return ICICLE_INVALID_ARGUMENT; // not a 2^k-th root of unity
    // the table holds 2^log_max elements: a root of an order no device memory can hold (stark252 has two-adicity 192, the
    // shift below would be undefined from 64 on) is refused here, cleanly, like the reference's failed allocation
    if (log_max > 36) return ICICLE_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)cfg->stream;
    const size_t n = (size_t)1 << log_max;
    uint32_t* tw = nullptr;
    HIP_TRY(hipMalloc(&tw, n * PR::NL32 * 4), ICICLE_ALLOCATION_FAILED);
    k_big_gen_twiddles<PR><<<(unsigned)((n / 64 + 256) / 256), 256, 0, st>>>(tw, mont_words<PR>(r), n);
    LAUNCH_CHECK("k_big_gen_twiddles", st);
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED); // see ntt.hip: published to every stream at once
    dom.tw = tw;
    dom.log_max = log_max;
    memset(dom.root, 0, sizeof dom.root);
    memcpy(dom.root, root, PR::NL32 * 4);
    return ICICLE_SUCCESS;
  }

  template <class PR>
  static icicle_error_t big_release_domain_run()
  {
    ICICLE_TRY(bind_current_device());
    const int dev = current_device_id();
    std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
    auto& m = BigDomainStore<PR>::map();
    for (auto it = m.begin(); it != m.end();) { // this device's domain + the copies its multi-device calls made elsewhere
      if (it->first != dev && it->second.owner != dev) {
        ++it;
        continue;
      }
      if (it->second.tw && hipSetDevice(it->first) == hipSuccess) {
        (void)hipDeviceSynchronize();
        (void)hipFree(it->second.tw);
        arena_trim(it->first); // the NTT work buffers cached for this domain's sizes go with it
      }
      it = m.erase(it);
    }
    (void)hipSetDevice(dev);
    return ICICLE_SUCCESS;
  }

  template <class PR>
  static icicle_error_t big_rou_from_domain_run(uint64_t logn, uint32_t* rou)
  {
    using F = FieldOps<PR>;
    if (!rou) return ICICLE_INVALID_POINTER;
    const int dev = current_device_id();
    std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
    auto it = BigDomainStore<PR>::map().find(dev);
    if (it == BigDomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT;
    if ((int64_t)logn > it->second.log_max) return ICICLE_INVALID_ARGUMENT;
    typename F::fe x = F::from_canonical(it->second.root); // twiddles[max >> logn] (cpu_ntt_domain.h:644-654)
    for (int i = 0; i < it->second.log_max - (int)logn; i++)
      x = F::sqr(x);
    F::to_canonical(rou, x);
    return ICICLE_SUCCESS;
  }

  template <class PR>
  static icicle_error_t big_get_root_of_unity_run(uint64_t max_size, uint32_t* rou)
  {
    using F = FieldOps<PR>;
    if (!rou || max_size == 0) return ICICLE_INVALID_ARGUMENT;
    uint32_t logn = 0;
    while (logn < 64 && ((uint64_t)1 << logn) < max_size)
      logn++; // ceil(log2(max_size)), src/ntt.cpp:57
    if ((int)logn > PR::TWO_ADICITY) return ICICLE_INVALID_ARGUMENT;
    typename F::fe x = F::from_canonical(PR::ROU32);
    for (int i = 0; i < PR::TWO_ADICITY - (int)logn; i++)
      x = F::sqr(x);
    if (logn == 0) x = F::one();
    F::to_canonical(rou, x);
    return ICICLE_SUCCESS;
  }

  // lanes > 1: every element is `lanes` field elements (an extension field: the transform runs per component with the
  // base field's twiddles)
  template <class PR>
  static icicle_error_t big_ntt_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u256_t* cfg, uint32_t* output, int lanes = 1);

  // row shards over device slots / row groups of a host-resident batch: see ntt_multi.hpp (same contract as ntt.hip)
  template <class PR>
  static icicle_error_t big_ntt_multi_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u256_t* cfg, uint32_t* output, int G, int max_slots, int lanes)
  {
    if (size <= 0 || !input || !output) return ICICLE_INVALID_ARGUMENT;
    ICICLE_TRY(bind_current_device());
    const int home = current_device_id();
    uint32_t root[8];
    {
      std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
      auto it = BigDomainStore<PR>::map().find(home);
      if (it == BigDomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT; // domain not initialised
      memcpy(root, it->second.root, 32);
    }
    icicle_ntt_config_u256_t sub = *cfg;
    sub.ext = nullptr;
    sub.are_inputs_on_device = sub.are_outputs_on_device = true;
    sub.is_async = true;
    NttRowsJob job;
    job.input = input, job.output = output;
    job.row_bytes = (size_t)size * PR::NL32 * 4 * lanes;
    job.batch = std::max(1, cfg->batch_size);
    job.G = G, job.max_slots = max_slots;
    job.in_on_device = cfg->are_inputs_on_device, job.out_on_device = cfg->are_outputs_on_device, job.is_async = cfg->is_async;
    job.stream = (hipStream_t)cfg->stream;
    return ntt_rows_multi(
      job,
      [&](const void* src, void* dst, int rows, hipStream_t st) -> icicle_error_t {
        icicle_ntt_config_u256_t c2 = sub;
        c2.stream = st;
        c2.batch_size = rows;
        return big_ntt_run<PR>((const uint32_t*)src, size, dir, &c2, (uint32_t*)dst, lanes);
      },
      [&](hipStream_t st) -> icicle_error_t {
        icicle_ntt_init_domain_config_t ic{st, false, nullptr};
        ICICLE_TRY(big_init_domain_run<PR>(root, &ic));
        std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
        auto& d = BigDomainStore<PR>::map()[current_device_id()];
        // "already initialised" is a silent success: a peer that holds a domain of a DIFFERENT root would transform its
        // row shards with other twiddles
        if (memcmp(d.root, root, PR::NL32 * 4) != 0) return ICICLE_INVALID_ARGUMENT;
        if (d.owner < 0 && current_device_id() != home) d.owner = home;
        return ICICLE_SUCCESS;
      });
  }

  template <class PR>
  static icicle_error_t big_ntt_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u256_t* cfg, uint32_t* output, int lanes)
  {
    using F = FieldOps<PR>;
    constexpr size_t EB = (size_t)PR::NL32 * 4; // bytes per field element
    using fe = typename F::fe;
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (cfg->ext && !cfg->columns_batch) {
      const int G = reinterpret_cast<const ConfigExt*>(cfg->ext)->get_int("hip_num_devices", 0);
      if (G >= 1) return big_ntt_multi_run<PR>(input, size, dir, cfg, output, G, 0, lanes);
    }
    if (!cfg->columns_batch && (!cfg->are_inputs_on_device || !cfg->are_outputs_on_device) && size > 0 && input && output && virtual_device_slots() == 0) {
      const int groups = ntt_host_row_groups((size_t)size * EB * lanes, std::max(1, cfg->batch_size));
      if (groups > 1) return big_ntt_multi_run<PR>(input, size, dir, cfg, output, groups, 1, lanes);
    }
    if (size <= 0 || (size & (size - 1)) != 0) return ICICLE_INVALID_ARGUMENT; // cpu_ntt_main.h:38-41
    if (!input || !output) return ICICLE_INVALID_POINTER;
    if (dir != ICICLE_NTT_FORWARD && dir != ICICLE_NTT_INVERSE) return ICICLE_INVALID_ARGUMENT;
    if (cfg->ordering < 0 || cfg->ordering > ICICLE_kMN) return ICICLE_INVALID_ARGUMENT;
    const int batch = std::max(1, cfg->batch_size);
    ICICLE_TRY(bind_current_device());
    const int dev = current_device_id();
    BigDomain dom;
    {
      std::lock_guard<std::mutex> g(BigDomainStore<PR>::mtx());
      auto it = BigDomainStore<PR>::map().find(dev);
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
    const size_t bytes = (size_t)n * batch * lanes * EB;

    TempBuf d_in_tmp, d_out_tmp, d_pw, d_work;
    const uint32_t* d_in = input;
    uint32_t* d_out = output;
    if (!cfg->are_inputs_on_device) {
      HIP_TRY(d_in_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_in_tmp.ptr(), input, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_in = d_in_tmp.as<uint32_t>();
    }
    if (!cfg->are_outputs_on_device) {
      if (!cfg->are_inputs_on_device) {
        d_out = d_in_tmp.as<uint32_t>(); // staged input doubles as the output buffer
      } else {
        HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
        d_out = d_out_tmp.as<uint32_t>();
      }
    }
    auto finish = [&]() -> icicle_error_t {
      if (!cfg->are_outputs_on_device) {
        HIP_TRY(hipMemcpyAsync(output, d_out, bytes, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
        HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
      } else if (!cfg->is_async) {
        HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
      }
      return ICICLE_SUCCESS;
    };
    if (logn == 0) { // size-1 transforms are the identity in every mode
      if (d_out != d_in) HIP_TRY(hipMemcpyAsync(d_out, d_in, bytes, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
      return finish();
    }

    NttLaunch nl;
    nl.logn = logn;
    nl.n = n;
    nl.nbatch = (uint32_t)(batch * lanes); // lane-transforms; offset(b') = (b'/lanes)*bs + (b'%lanes), in field elements
    nl.lanes = (uint32_t)lanes;
    if (cfg->columns_batch) { // element j of transform b at j*batch + b (ntt_cpu.h:250,274-275)
      nl.bs = (uint64_t)lanes;
      nl.es = (uint64_t)batch * lanes;
    } else {
      nl.bs = n * lanes;
      nl.es = (uint64_t)lanes;
    }
    const int ord = cfg->ordering;
    nl.in_rev = (ord == ICICLE_kRN || ord == ICICLE_kRR);
    nl.out_rev = (ord == ICICLE_kNR || ord == ICICLE_kRR);
    nl.inverse = (dir == ICICLE_NTT_INVERSE);
    nl.log_max = dom.log_max;
    nl.ninv_mont = 0;
    BigWords ninv{};
    if (nl.inverse) ninv = mont_words<PR>(host_ninv<PR>(logn));
    nl.coset = !words_is_one(cfg->coset_gen);
    if (nl.coset) {
      HIP_TRY(d_pw.alloc(n * EB, st), ICICLE_ALLOCATION_FAILED);
      fe gm = F::from_canonical(cfg->coset_gen);
      if (nl.inverse) gm = host_inverse<PR>(gm);
      k_big_coset_powers<PR><<<(unsigned)((n / 16 + 256) / 256), 256, 0, st>>>(d_pw.as<uint32_t>(), mont_words<PR>(gm), n);
      LAUNCH_CHECK("k_big_coset_powers", st);
    }

    int parts[3], P;
    split_logn(logn, 8, parts, &P);
    uint32_t* Wk = nullptr;
    if (P >= 2) { // see ntt.hip: the last pass permutes across tiles, earlier passes run in a work buffer
      HIP_TRY(d_work.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      Wk = d_work.as<uint32_t>();
    }
    HIP_TRY(hipFuncSetAttribute((const void*)k_big_ntt_pass<PR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), ICICLE_INVALID_ARGUMENT);
    for (int p = 0; p < P; p++) {
      const uint32_t* src = (p == 0) ? d_in : Wk;
      uint32_t* dst = (p == P - 1) ? d_out : Wk;
      const uint64_t L = (uint64_t)1 << parts[p];
      // tile of L x T elements (36 B each in LDS for the 256-bit fields), <= 72 KiB so that two blocks share a CU; runs
      // of T elements are >= 128 B in HBM: T = 4 of 32 B, 16 of 8 B
      uint32_t tmax = std::max(4, 32 / PR::NL32);
      while (tmax > 1 && L * tmax * sizeof(fe) > 72 * 1024)
        tmax >>= 1;
      const PassDesc pd = make_pass(parts, P, p, n, dom.log_max, tmax);
      const uint32_t tot = (uint32_t)(L * pd.T);
      // one thread per 4-element group of a stage pair; goldilocks: per 16-element group of four stages (when the pass has four)
      const unsigned threads = std::max(64u, std::min(512u, tot / ((PR::NL32 == 2 && parts[p] >= 4) ? 16 : 4)));
      for (uint32_t r0 = 0; r0 < nl.nbatch; r0 += 65535) {
        NttLaunch ns = nl;
        ns.row0 = r0;
        k_big_ntt_pass<PR><<<dim3(pd.ntiles, std::min<uint32_t>(65535, nl.nbatch - r0)), threads, (size_t)tot * sizeof(fe), st>>>(src, dst, dom.tw, d_pw.as<uint32_t>(), pd, ns, ninv);
      }
      LAUNCH_CHECK("k_big_ntt_pass", st);
    }
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);
    return finish();
  }

} // namespace icicle_hip

using namespace icicle_hip;

#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

// domain / root-of-unity entry points (src/ntt.cpp:26,41,55,75) + collision-free aliases for the plugin
#define DEFINE_NTT_DOMAIN_API(F, PRM)                                                                                  \
  extern "C" icicle_error_t F##_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config) \
  {                                                                                                                    \
    GUARDED(big_init_domain_run<PRM>(primitive_root, config));                                                         \
  }                                                                                                                    \
  extern "C" icicle_error_t F##_ntt_release_domain(void) { GUARDED(big_release_domain_run<PRM>()); }                   \
  extern "C" icicle_error_t F##_get_root_of_unity(uint64_t max_size, uint32_t* rou)                                    \
  {                                                                                                                    \
    GUARDED(big_get_root_of_unity_run<PRM>(max_size, rou));                                                            \
  }                                                                                                                    \
  extern "C" icicle_error_t F##_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou)                            \
  {                                                                                                                    \
    GUARDED(big_rou_from_domain_run<PRM>(logn, rou));                                                                  \
  }                                                                                                                    \
  extern "C" icicle_error_t icicle_hip_##F##_ntt_init_domain(const uint32_t* r, const icicle_ntt_init_domain_config_t* c) { GUARDED(big_init_domain_run<PRM>(r, c)); } \
  extern "C" icicle_error_t icicle_hip_##F##_ntt_release_domain(void) { GUARDED(big_release_domain_run<PRM>()); }      \
  extern "C" icicle_error_t icicle_hip_##F##_get_root_of_unity_from_domain(uint64_t l, uint32_t* r) { GUARDED(big_rou_from_domain_run<PRM>(l, r)); }

#define DEFINE_NTT_U256(F, PRM)                                                                                        \
  extern "C" icicle_error_t F##_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u256_t* config, uint32_t* output) \
  {                                                                                                                    \
    GUARDED(big_ntt_run<PRM>(input, size, dir, config, output));                                                       \
  }                                                                                                                    \
  extern "C" icicle_error_t icicle_hip_##F##_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u256_t* c, uint32_t* o) { GUARDED(big_ntt_run<PRM>(i, n, d, c, o)); } \
  DEFINE_NTT_DOMAIN_API(F, PRM)

DEFINE_NTT_U256(bn254, bn254_fr_params)
DEFINE_NTT_U256(bls12_381, bls12_381_fr_params)
DEFINE_NTT_U256(bls12_377, bls12_377_fr_params)
DEFINE_NTT_U256(stark252, stark252_fr_params) // reference FIELD_ID 1002: a 252-bit field with an NTT and no curve


// Goldilocks (reference FIELD_ID 1005: NTT + EXT_FIELD): 8-byte elements, NTTConfig<scalar_t> is 40 bytes with a 2-word
// coset generator; the quadratic extension field (u^2 = 7) transforms its two components with the base field's twiddles.
static icicle_error_t goldilocks_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u64_t* c, uint32_t* output, int lanes)
{
  if (!c) return ICICLE_INVALID_POINTER;
  icicle_ntt_config_u256_t w{};
  w.stream = c->stream;
  w.coset_gen[0] = c->coset_gen[0], w.coset_gen[1] = c->coset_gen[1];
  w.batch_size = c->batch_size, w.columns_batch = c->columns_batch, w.ordering = c->ordering;
  w.are_inputs_on_device = c->are_inputs_on_device, w.are_outputs_on_device = c->are_outputs_on_device, w.is_async = c->is_async;
  w.ext = c->ext;
  return big_ntt_run<goldilocks_params>(input, size, dir, &w, output, lanes);
}
extern "C" icicle_error_t goldilocks_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u64_t* c, uint32_t* o) { GUARDED(goldilocks_run(i, n, d, c, o, 1)); }
extern "C" icicle_error_t goldilocks_extension_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u64_t* c, uint32_t* o) { GUARDED(goldilocks_run(i, n, d, c, o, 2)); }
extern "C" icicle_error_t icicle_hip_goldilocks_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u64_t* c, uint32_t* o) { GUARDED(goldilocks_run(i, n, d, c, o, 1)); }
extern "C" icicle_error_t icicle_hip_goldilocks_extension_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u64_t* c, uint32_t* o) { GUARDED(goldilocks_run(i, n, d, c, o, 2)); }
DEFINE_NTT_DOMAIN_API(goldilocks, goldilocks_params)
