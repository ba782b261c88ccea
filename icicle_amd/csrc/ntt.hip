// NTT for gfx950 over 31-bit fields (BabyBear, KoalaBear).
//
// Reference semantics: icicle/backend/cpu/include/ntt_cpu.h:70-232 (run), :247-306 (input
// reorder), :317-364 (coset), cpu_ntt_domain.h:65-110,614-654 (domain), ntt_task.h:1208-1236
// (radix-2 DIT, inverse twiddle = table[max - idx], 1/N folded into the last layer).
// SURVEY.md App. C lists the fine print (orderings, coset, batch layouts, kNM/kMN = natural).
//
// Design (not the CPU's Winograd-leaf hierarchy): N = N0*N1[*N2] is split into at most three
// passes of 2^8-point (above 2^24: up to 2^10-point) sub-transforms. A pass stages a [2^s x T] tile (T adjacent columns, so
// every HBM access is a run of T contiguous elements) in LDS, runs the s radix-2 stages there,
// applies the inter-pass twiddle w_M^(j*K) on the way out, and the LAST pass scatters straight
// into natural order (runs of T again) -- so an NN transform is exactly P reads + P writes of the
// data with no separate transpose / bit-reverse pass. Data stays in canonical form throughout:
// twiddles are stored in Montgomery form, and montmul(canonical, w*R) = canonical.
#include "common.h"
#include "smallfield.hpp"
#include "ntt_plan.h"
#include "ntt_fast.hpp" // k_ntt_fast: the register-blocked pass kernel
#include "ntt_multi.hpp"
#include <thread>
#include <algorithm>
#include <cmath>

namespace icicle_hip {

  // ---- per-device, per-field twiddle domain (reference: one global domain per device) ----------
  struct NttDomain {
    uint32_t* tw = nullptr; // tw[i] = w_max^i (Montgomery), i < max_size
    int log_max = -1;
    uint32_t root = 0; // canonical w_max
    int owner = -1;    // >= 0: brought up by a multi-device call of device `owner`, released with that device's domain
  };
  template <class PR>
  struct DomainStore {
    static std::mutex& mtx()
    {
      static std::mutex m;
      return m;
    }
    static std::map<int, NttDomain>& map()
    {
      static std::map<int, NttDomain> m;
      return m;
    }
  };

  template <class PR>
  __global__ __launch_bounds__(256) void k_gen_twiddles(uint32_t* __restrict__ tw, uint32_t root_mont, size_t n)
  {
    using S = SmallField<PR>;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) tw[i] = S::pow(root_mont, (uint64_t)i);
  }

  // ---- generic pass (any ordering / coset): one radix-2 stage per LDS round trip ---------------
  // Correctness-first kernel used when the input or output is bit-reversed (kRN/kNR/kRR) or a coset
  // generator is given; the headline path (kNN, no coset) uses k_ntt_fast below.
  // grid = (ntiles, nbatch). Dynamic LDS: L*T u32.
  template <class PR>
  __global__ __launch_bounds__(1024) void k_ntt_pass_generic(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ tw, const uint32_t* __restrict__ coset_pow, PassDesc pd, NttLaunch nl)
  {
    using S = SmallField<PR>;
    extern __shared__ uint32_t tile[];
    const uint32_t L = 1u << pd.s, T = pd.T;
    const uint32_t bprime = nl.row0 + blockIdx.y; // launched in slices of <= 65535 rows
    const uint64_t boff = (uint64_t)(bprime / nl.lanes) * nl.bs + (bprime % nl.lanes);
    const uint32_t a = blockIdx.x / pd.tiles_per_a, ct = blockIdx.x % pd.tiles_per_a;
    const uint64_t in_base = (uint64_t)a * pd.in_base_a + (uint64_t)ct * pd.in_base_ct;
    const uint64_t max_mask = ((uint64_t)1 << nl.log_max) - 1;
    const uint32_t tot = L * T;

    // ---- load: logical (k,t) -> LDS row bitrev_s(k)
    for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {
      const uint32_t t = e % T, k = e / T;
      uint64_t addr = in_base + (uint64_t)k * pd.in_sk + (uint64_t)t * pd.in_st; // logical index in the row
      uint64_t maddr = nl.in_rev && pd.pidx == 0 ? bitrev64(addr, nl.logn) : addr;
      uint32_t v = in[boff + maddr * nl.es];
      if (nl.coset && !nl.inverse && pd.pidx == 0) v = S::mul(v, coset_pow[addr]);
      tile[(__brev(k) >> (32 - pd.s)) * T + t] = v;
    }
    __syncthreads();

    // ---- s radix-2 DIT stages in LDS; w_L^e = tw[e * (max/L)], inverse uses tw[max - idx]
    const uint32_t lstride_log = nl.log_max - pd.s;
    for (int q = 0; q < pd.s; q++) {
      const uint32_t half = 1u << q;
      for (uint32_t id = threadIdx.x; id < tot / 2; id += blockDim.x) {
        const uint32_t t = id % T, bf = id / T;
        const uint32_t pos = bf & (half - 1);
        const uint32_t i = ((bf >> q) << (q + 1)) + pos;
        uint64_t widx = ((uint64_t)pos << (pd.s - 1 - q)) << lstride_log;
        if (nl.inverse) widx = (((uint64_t)1 << nl.log_max) - widx) & max_mask;
        const uint32_t w = tw[widx];
        const uint32_t u = tile[i * T + t];
        const uint32_t v = S::mul(tile[(i + half) * T + t], w);
        tile[i * T + t] = S::add(u, v);
        tile[(i + half) * T + t] = S::sub(u, v);
      }
      __syncthreads();
    }

    // ---- store
    for (uint32_t e = threadIdx.x; e < tot; e += blockDim.x) {
      const uint32_t t = e % T, k = e / T;
      uint32_t v = tile[k * T + t];
      uint64_t oaddr;
      if (!pd.is_last) {
        // in place: same slot as the load; multiply by w_M^(jnext*K)
        oaddr = in_base + (uint64_t)k * pd.in_sk + (uint64_t)t * pd.in_st;
        const uint64_t c = (uint64_t)ct * T + t;            // column index within [0,C)
        const uint64_t jnext = c / pd.cprime;
        const uint64_t K = (pd.pidx == 0) ? (uint64_t)k : ((uint64_t)a + (uint64_t)pd.n0 * k);
        uint64_t widx = (jnext * K * pd.tw_stride) & max_mask;
        if (nl.inverse) widx = (((uint64_t)1 << nl.log_max) - widx) & max_mask;
        v = S::mul(v, tw[widx]);
      } else {
        // natural-order scatter: K = digitrev(a_t) + (N/L)*k
        uint64_t K0;
        if (pd.pidx <= 1) {
          K0 = (uint64_t)ct * T + t;                        // P=1: 0 ; P=2: a = k0
        } else {
          const uint64_t k0 = (uint64_t)ct * T + t, k1 = a; // P=3: tile over k0, a carries k1
          K0 = k0 + (uint64_t)pd.n0 * k1;
        }
        oaddr = K0 + (uint64_t)k * pd.out_sk;
        if (nl.inverse) {
          v = S::mul(v, nl.ninv_mont);
          if (nl.coset) v = S::mul(v, coset_pow[oaddr]);
        }
        if (nl.out_rev) oaddr = bitrev64(oaddr, nl.logn);
      }
      out[boff + oaddr * nl.es] = v;
    }
  }



  // ---- bit-reversed input (kRN / kRR): explicit reordering pre-pass ---------------------------------
  // Reading x[bitrev(j)] inside pass 0 would turn its 128-byte runs into 4-byte gathers, so the rows are
  // bit-reversed once into the work buffer (one extra read + write of the data, ~1/3 of a transform) and the
  // natural-order passes run from there. Row-major rows: 32 x 32 tiles through LDS, index = hi5 | mid | lo5,
  // read along lo5, written along bitrev(hi5) -- both sides in 128-byte runs.
  __global__ __launch_bounds__(256) void k_bitrev_rows_tiled(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t logn, uint64_t bs)
  {
    __shared__ uint32_t tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5; // 256 threads, 4 rows each
    const uint32_t midbits = logn - 10;
    const uint64_t mid = blockIdx.x;
    const uint64_t base = (uint64_t)blockIdx.y * bs;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t ty = ty0 + 8 * i;
      v[i] = in[base + (((uint64_t)ty << (logn - 5)) | (mid << 5) | tx)];
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      tile[ty0 + 8 * i][tx] = v[i];
    __syncthreads();
    const uint64_t rmid = bitrev64(mid, midbits);
    const uint32_t hi = __brev(tx) >> 27;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t ty = ty0 + 8 * i, lo = __brev(ty) >> 27;
      out[base + (((uint64_t)ty << (logn - 5)) | (rmid << 5) | tx)] = tile[hi][lo];
    }
  }
  // any layout: one thread per (element, transform), transforms fastest (coalesced for columns_batch)
  __global__ __launch_bounds__(256) void k_bitrev_rows_simple(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t logn, uint32_t nbatch, uint32_t lanes, uint64_t bs, uint64_t es, bool batch_fastest)
  {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)1 << logn;
    if (t >= n * nbatch) return;
    const uint64_t j = batch_fastest ? t / nbatch : t % n;
    const uint32_t b = (uint32_t)(batch_fastest ? t % nbatch : t / n);
    const uint64_t boff = (uint64_t)(b / lanes) * bs + (b % lanes);
    out[boff + bitrev64(j, logn) * es] = in[boff + j * es];
  }

  // two-level coset table for the fast path: ctab[i] = g^i (i < 4096), ctab[4096 + i] = scale * g^(4096 i);
  // g is the generator (forward) or its inverse (inverse transform, scale = 1/N so one product applies both)
  template <class PR>
  __global__ __launch_bounds__(256) void k_coset_tables(uint32_t* __restrict__ ctab, uint32_t g_mont, uint32_t scale_mont, uint32_t nhi)
  {
    using S = SmallField<PR>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4096) ctab[i] = S::pow(g_mont, (uint64_t)i);
    if (i < nhi) ctab[4096 + i] = S::mul(scale_mont, S::pow(g_mont, (uint64_t)i << 12));
  }

  // coset powers: pw[j] = g^j (forward) or g^-j (inverse), Montgomery
  template <class PR>
  __global__ __launch_bounds__(256) void k_coset_powers(uint32_t* __restrict__ pw, uint32_t g_mont, uint64_t n)
  {
    using S = SmallField<PR>;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t j0 = t * 16;
    if (j0 >= n) return;
    uint32_t v = S::pow(g_mont, j0);
    for (int q = 0; q < 16 && j0 + q < n; q++) {
      pw[j0 + q] = v;
      v = S::mul(v, g_mont);
    }
  }

  // data[r][c] *= w_N^(+-(row0 + r) * c): the inter-step twiddle of a 4-step transform whose two steps run
  // on different GPUs (icicle_amd/dist.py ntt_distributed); w_N = tw[max/N]
  template <class PR>
  __global__ __launch_bounds__(256) void k_twiddle_rows(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t logn_total, uint32_t log_max, int inverse)
  {
    using S = SmallField<PR>;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const uint64_t r = t / cols, c = t - r * cols;
    const uint64_t nmask = ((uint64_t)1 << logn_total) - 1;
    const uint64_t e = ((row0 + r) * c) & nmask; // exponent mod N
    uint64_t idx = e << (log_max - logn_total);
    const uint64_t max_mask = ((uint64_t)1 << log_max) - 1;
    if (inverse) idx = (((uint64_t)1 << log_max) - idx) & max_mask;
    data[t] = S::mul(data[t], tw[idx]);
  }

  template <class PR>
  static icicle_error_t twiddle_rows_run(uint32_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t logn_total, bool inverse, hipStream_t st)
  {
    if (!data) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    NttDomain dom;
    {
      std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
      auto it = DomainStore<PR>::map().find(current_device_id());
      if (it == DomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT;
      dom = it->second;
    }
    if ((int)logn_total > dom.log_max) return ICICLE_INVALID_ARGUMENT;
    const uint64_t tot = rows * cols;
    if (tot == 0) return ICICLE_SUCCESS;
    k_twiddle_rows<PR><<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(data, dom.tw, rows, cols, row0, logn_total, (uint32_t)dom.log_max, inverse ? 1 : 0);
    LAUNCH_CHECK("k_twiddle_rows", st);
    return ICICLE_SUCCESS;
  }

  // n == 1 and pure-copy helper with strides
  __global__ void k_copy_strided(const uint32_t* in, uint32_t* out, uint64_t count)
  {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < count) out[i] = in[i];
  }

  template <class PR>
  static icicle_error_t ntt_init_domain_run(const uint32_t* root, const icicle_ntt_init_domain_config_t* cfg)
  {
    using S = SmallField<PR>;
    if (!root || !cfg) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    const int dev = current_device_id();
    std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
    auto& dom = DomainStore<PR>::map()[dev];
    if (dom.tw) return ICICLE_SUCCESS; // already initialised: silent success (cpu_ntt_domain.h:69)
    const uint32_t r = *root;
    if (r == 0 || r >= PR::P) return ICICLE_INVALID_ARGUMENT;
    // order of the root by repeated squaring (cpu_ntt_domain.h:78-94)
    uint32_t x = S::to_mont(r);
    int log_max = 0;
    while (x != S::one() && log_max <= PR::TWO_ADICITY) {
      x = S::mul(x, x);
      log_max++;
    }
    if (x != S::one()) return ICICLE_INVALID_ARGUMENT; // not a 2^k-th root of unity
    if (log_max > 36) return ICICLE_INVALID_ARGUMENT;  // (same guard as ntt_big.hip; 31-bit fields stop at 2^27)
    hipStream_t st = (hipStream_t)cfg->stream;
    const size_t n = (size_t)1 << log_max;
    uint32_t* tw = nullptr;
    HIP_TRY(hipMalloc(&tw, n * 4), ICICLE_ALLOCATION_FAILED);
    k_gen_twiddles<PR><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(tw, S::to_mont(r), n);
    LAUNCH_CHECK("k_gen_twiddles", st);
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);
    // always synchronous: the table is published to every stream of this device the moment init returns (an ntt()
    // on another stream must never read it half-written); domain init is a one-time operation
    HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    dom.tw = tw;
    dom.log_max = log_max;
    dom.root = r;
    return ICICLE_SUCCESS;
  }

  template <class PR>
  static icicle_error_t ntt_release_domain_run()
  {
    ICICLE_TRY(bind_current_device());
    const int dev = current_device_id();
    std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
    auto& m = DomainStore<PR>::map();
    // this device's domain and the copies a multi-device call of this device brought up elsewhere
    for (auto it = m.begin(); it != m.end();) {
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
  static icicle_error_t ntt_rou_from_domain_run(uint64_t logn, uint32_t* rou)
  {
    using S = SmallField<PR>;
    if (!rou) return ICICLE_INVALID_POINTER;
    const int dev = current_device_id();
    std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
    auto it = DomainStore<PR>::map().find(dev);
    if (it == DomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT;
    if ((int64_t)logn > it->second.log_max) return ICICLE_INVALID_ARGUMENT;
    // twiddles[max >> logn] (cpu_ntt_domain.h:644-654)
    uint32_t x = S::to_mont(it->second.root);
    for (int i = 0; i < it->second.log_max - (int)logn; i++)
      x = S::mul(x, x);
    *rou = S::from_mont(x);
    return ICICLE_SUCCESS;
  }

  template <class PR>
  static icicle_error_t get_root_of_unity_run(uint64_t max_size, uint32_t* rou)
  {
    using S = SmallField<PR>;
    if (!rou || max_size == 0) return ICICLE_INVALID_ARGUMENT;
    uint32_t logn = 0;
    while (logn < 64 && ((uint64_t)1 << logn) < max_size)
      logn++; // ceil(log2(max_size)), src/ntt.cpp:57
    if ((int)logn > PR::TWO_ADICITY) return ICICLE_INVALID_ARGUMENT;
    uint32_t x = S::to_mont(PR::ROU);
    for (int i = 0; i < PR::TWO_ADICITY - (int)logn; i++)
      x = S::mul(x, x);
    *rou = logn == 0 ? 1u : S::from_mont(x);
    return ICICLE_SUCCESS;
  }

  // k_ntt_fast variant for a pass of 2^s points (ntt_fast.hpp: pick_pass_t); the lane-native set (interleaved transforms)
  // is instantiated in ntt_lanes.hip
  template <class PR>
  static pass_fn_t pick_pass(int s, bool dif, bool inv, bool coset, bool outrev, bool v4, bool big = false)
  {
    if (big && !dif && !coset && !outrev && s == 9) return (pass_fn_t)k_ntt_fast<PR, 1, 3, false, false, false, false, false, true>;
    if (big && !dif && !coset && !outrev && s == 10) return (pass_fn_t)k_ntt_fast<PR, 2, 3, false, false, false, false, false, true>;
    if (big) return nullptr;
    if (v4 && dif && !coset && !outrev && s == 8) // 16-byte lanes: row pass of 2^8 sub-transforms, 32-column tiles
      return inv ? (pass_fn_t)k_ntt_fast<PR, 4, 2, true, true, false, false, true> : (pass_fn_t)k_ntt_fast<PR, 4, 2, true, false, false, false, true>;
    return pick_pass_t<PR, false>(s, dif, inv, coset, outrev);
  }

  template <class PR>
  static icicle_error_t ntt_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* cfg, uint32_t* output, uint32_t lanes);

#include "ntt_split.hpp"

  // Batched NTT over several devices behind the unchanged <field>_ntt symbol: config.ext {"hip_num_devices": G} cuts
  // the batch into G contiguous row shards (rows are independent transforms: no collective) over
  // min(G, visible GPUs) device slots -- ntt_multi.hpp holds the slot threads and the upload / compute / download
  // pipeline. The twiddle domain is brought up on a device the first time it is used (same root as the calling
  // device's) and released together with the calling device's domain. Row-major batches only.
  template <class PR>
  static icicle_error_t ntt_multi_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* cfg, uint32_t* output, uint32_t lanes, int G, int max_slots)
  {
    if (size <= 0 || !input || !output) return ICICLE_INVALID_ARGUMENT;
    ICICLE_TRY(bind_current_device());
    const int home = current_device_id();
    uint32_t root = 0;
    {
      std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
      auto it = DomainStore<PR>::map().find(home);
      if (it == DomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT; // domain not initialised
      root = it->second.root;
    }
    icicle_ntt_config_u32_t sub = *cfg;
    sub.ext = nullptr;
    sub.are_inputs_on_device = sub.are_outputs_on_device = true;
    sub.is_async = true;
    NttRowsJob job;
    job.input = input, job.output = output;
    job.row_bytes = (size_t)size * lanes * 4;
    job.batch = std::max(1, cfg->batch_size);
    job.G = G, job.max_slots = max_slots;
    job.in_on_device = cfg->are_inputs_on_device, job.out_on_device = cfg->are_outputs_on_device, job.is_async = cfg->is_async;
    job.stream = (hipStream_t)cfg->stream;
    return ntt_rows_multi(
      job,
      [&](const void* src, void* dst, int rows, hipStream_t st) -> icicle_error_t {
        icicle_ntt_config_u32_t c2 = sub;
        c2.stream = st;
        c2.batch_size = rows;
        return ntt_run<PR>((const uint32_t*)src, size, dir, &c2, (uint32_t*)dst, lanes);
      },
      [&](hipStream_t st) -> icicle_error_t { // silent success if this device already has the domain
        icicle_ntt_init_domain_config_t ic{st, false, nullptr};
        ICICLE_TRY(ntt_init_domain_run<PR>(&root, &ic));
        std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
        auto& d = DomainStore<PR>::map()[current_device_id()];
        if (d.root != root) return ICICLE_INVALID_ARGUMENT; // a peer with a domain of another root: other twiddles
        if (d.owner < 0 && current_device_id() != home) d.owner = home; // brought up on behalf of `home`: goes when home's domain goes
        return ICICLE_SUCCESS;
      });
  }

  template <class PR>
  static icicle_error_t ntt_run(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* cfg, uint32_t* output, uint32_t lanes)
  {
    using S = SmallField<PR>;
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (cfg->ext && !cfg->columns_batch) {
      const int G = reinterpret_cast<const ConfigExt*>(cfg->ext)->get_int("hip_num_devices", 0);
      // "hip_force_rccl" with ONE device: the split transform's exchanges as sends to the own rank (the real librccl on a 1-GPU box)
      const bool force = reinterpret_cast<const ConfigExt*>(cfg->ext)->get_bool("hip_force_rccl", false) && G == 1;
      if (G >= 1) {
        // fewer transforms than devices: cut every transform itself over the device slots (four-step, all-to-all
        // exchanges: ntt_split.hpp) when its shape allows; otherwise (and for every batch >= G) rows over devices
        if ((std::max(1, cfg->batch_size) < G || force) && lanes == 1 && cfg->ordering == ICICLE_kNN && cfg->coset_gen == 1 && size > 0 && (size & (size - 1)) == 0 && input && output &&
            (dir == ICICLE_NTT_FORWARD || dir == ICICLE_NTT_INVERSE)) {
          DeviceSlots ds;
          ICICLE_TRY(make_device_slots(G, &ds));
          int logn = 0;
          while ((1 << logn) < size)
            logn++;
          SplitShape shp;
          if (ds.P == G && split_shape(logn, ds.P, &shp, force)) return ntt_split_run<PR>(input, size, dir, cfg, output, ds, shp, force);
        }
        return ntt_multi_run<PR>(input, size, dir, cfg, output, lanes, G, 0);
      }
    }
    // host-resident rows of a large batch on one GPU: row groups through the upload / compute / download pipeline
    if (!cfg->columns_batch && (!cfg->are_inputs_on_device || !cfg->are_outputs_on_device) && size > 0 && input && output && virtual_device_slots() == 0) {
      const int groups = ntt_host_row_groups((size_t)size * lanes * 4, std::max(1, cfg->batch_size));
      if (groups > 1) return ntt_multi_run<PR>(input, size, dir, cfg, output, lanes, groups, 1);
    }
    if (size <= 0 || (size & (size - 1)) != 0) return ICICLE_INVALID_ARGUMENT; // cpu_ntt_main.h:38-41
    if (!input || !output) return ICICLE_INVALID_POINTER;
    if (dir != ICICLE_NTT_FORWARD && dir != ICICLE_NTT_INVERSE) return ICICLE_INVALID_ARGUMENT;
    if (cfg->ordering < 0 || cfg->ordering > ICICLE_kMN) return ICICLE_INVALID_ARGUMENT;
    const int batch = std::max(1, cfg->batch_size);
    ICICLE_TRY(bind_current_device());
    const int dev = current_device_id();
    NttDomain dom;
    {
      std::lock_guard<std::mutex> g(DomainStore<PR>::mtx());
      auto it = DomainStore<PR>::map().find(dev);
      if (it == DomainStore<PR>::map().end() || !it->second.tw) return ICICLE_INVALID_ARGUMENT; // domain not initialised
      dom = it->second;
    }
    int logn = 0;
    while ((1 << logn) < size)
      logn++;
    if (logn > dom.log_max) return ICICLE_INVALID_ARGUMENT;
    if (cfg->coset_gen == 0 || cfg->coset_gen >= PR::P) return ICICLE_INVALID_ARGUMENT;

    hipStream_t st = (hipStream_t)cfg->stream;
    const uint64_t n = (uint64_t)size;
    const uint64_t total = n * batch * lanes;
    const size_t bytes = total * 4;

    TempBuf d_in_tmp, d_out_tmp, d_pw;
    const uint32_t* d_in = input;
    uint32_t* d_out = output;
    if (!cfg->are_inputs_on_device) {
      HIP_TRY(d_in_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_in_tmp.ptr(), input, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_in = d_in_tmp.as<uint32_t>();
    }
    if (!cfg->are_outputs_on_device) {
      // reuse the staged input buffer as output when both are host-side (in-place on device)
      if (!cfg->are_inputs_on_device) {
        d_out = d_in_tmp.as<uint32_t>();
      } else {
        HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
        d_out = d_out_tmp.as<uint32_t>();
      }
    }

    NttLaunch nl;
    nl.logn = logn;
    nl.n = n;
    nl.nbatch = (uint32_t)batch * lanes;
    nl.lanes = lanes;
    if (cfg->columns_batch) { // element j of transform b at (j*batch + b)*lanes + lane (ntt_cpu.h:250,274-275)
      nl.bs = lanes;
      nl.es = (uint64_t)batch * lanes;
    } else {
      nl.bs = n * lanes;
      nl.es = lanes;
    }
    const int ord = cfg->ordering;
    nl.in_rev = (ord == ICICLE_kRN || ord == ICICLE_kRR);
    nl.out_rev = (ord == ICICLE_kNR || ord == ICICLE_kRR);
    nl.inverse = (dir == ICICLE_NTT_INVERSE);
    nl.log_max = dom.log_max;
    {
      uint32_t two_inv = S::inv(S::to_mont(2));
      nl.ninv_mont = S::pow(two_inv, (uint64_t)logn);
    }
    nl.coset = (cfg->coset_gen != 1);
    uint32_t coset_g = S::one();
    if (nl.coset) {
      coset_g = S::to_mont(cfg->coset_gen);
      if (nl.inverse) coset_g = S::inv(coset_g);
    }

    if (logn == 0) { // size-1 transforms are the identity in every mode
      HIP_TRY(hipMemcpyAsync(d_out, d_in, bytes, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
      if (!cfg->are_outputs_on_device) {
        HIP_TRY(hipMemcpyAsync(output, d_out, bytes, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
        HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
      } else if (!cfg->is_async) {
        HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
      }
      return ICICLE_SUCCESS;
    }
    int parts[3], P;
    // sub-transforms of 2^8 points (three passes at 2^24). Two passes of 2^12 (1024-thread blocks, 4-column tiles)
    // and 64-column tiles were built and measured in round 1 and lost: profiles/r01_notes.md.
    split_logn(logn, 8, parts, &P);
    {
      // Which pass takes the extra bit of a size that is not a multiple of three: split_logn gives it to the first ones (9,8,8 at
      // 2^25). Measured (profiles/r05_ntt_big_parts.txt, ICICLE_HIP_NTT_PARTS_ORDER = 0 / 1 / 2 = first / last / middle): at 2^25 the
      // MIDDLE pass is the place for the 512-row sub-transform, 8,9,8: 6.48 -> 6.02 ms (x 32 rows); every other size measured
      // (2^20, 2^22, 2^26, 2^27) is best or within 3 % with the default, so only 2^25 changes.
      static const int order = getenv("ICICLE_HIP_NTT_PARTS_ORDER") ? atoi(getenv("ICICLE_HIP_NTT_PARTS_ORDER")) : -1;
      if (P == 3 && (order == 1)) std::swap(parts[0], parts[2]);
      if (P == 3 && (order == 2 || (order < 0 && logn == 25))) {
        std::swap(parts[0], parts[1]);
        if (parts[0] > parts[2]) std::swap(parts[0], parts[2]);
      }
    }
    // P >= 2: passes 0..P-2 run in a work buffer (the last pass permutes across tiles, so it can
    // never be in place; this also makes input == output legal, test_mod_arithmetic_api.h:627,679)
    TempBuf d_work;
    uint32_t* W = nullptr;

    // fast path: natural-order input (kNN / kNM / kMN / kNR), or bit-reversed input after the reordering pre-pass
    // below (kRN / kRR, P >= 2); cosets and single-pass reversed-input transforms use the generic kernel
    // kRN: the passes consume the bit-reversed rows as they lie (ntt_fast.hpp RN, in place behind pass 0); kRR and forward cosets
    // still reorder first. ICICLE_HIP_NTT_RN_NATIVE=0: the pre-pass for every reversed input (rounds 1-4).
    static const bool rn_on = !(getenv("ICICLE_HIP_NTT_RN_NATIVE") && atoi(getenv("ICICLE_HIP_NTT_RN_NATIVE")) == 0);
    const bool rn_native = rn_on && nl.in_rev && !nl.out_rev && !(nl.coset && !nl.inverse);
    const bool prerev = nl.in_rev && P >= 2 && !rn_native;
    const bool fast = !nl.in_rev || prerev || rn_native;
    TempBuf d_ctab;
    if (nl.coset && fast) { // two-level table, 4096 + N/4096 entries
      const uint32_t nhi = (uint32_t)std::max<uint64_t>(1, n >> 12);
      HIP_TRY(d_ctab.alloc((size_t)(4096 + nhi) * 4, st), ICICLE_ALLOCATION_FAILED);
      k_coset_tables<PR><<<(std::max(4096u, nhi) + 255) / 256, 256, 0, st>>>(d_ctab.as<uint32_t>(), coset_g, nl.inverse ? nl.ninv_mont : S::one(), nhi);
      LAUNCH_CHECK("k_coset_tables", st);
    } else if (nl.coset) { // generic kernel: full table of powers
      HIP_TRY(d_pw.alloc(n * 4, st), ICICLE_ALLOCATION_FAILED);
      k_coset_powers<PR><<<(unsigned)((n / 16 + 256) / 256), 256, 0, st>>>(d_pw.as<uint32_t>(), coset_g, n);
      LAUNCH_CHECK("k_coset_powers", st);
    }
    // Row groups (experimental, OFF by default: ICICLE_HIP_NTT_GROUP_MB=<MiB>): a group of rows runs
    // through ALL passes before the next group starts so that pass p+1 could read pass p's output from
    // the 256 MiB Infinity Cache. Measured on MI355X (profiles/r01_notes.md): 3.4x SLOWER at 64-192 MiB
    // groups, because a block then serves 1-2 rows and the per-block twiddle gathers (46 per thread)
    // are no longer amortised over the batch. Kept for experiments only.
    uint32_t rows_per_group = nl.nbatch;
    if (fast && !prerev && !rn_native && P >= 2 && !cfg->columns_batch && lanes == 1) {
      size_t group_mb = 0;
      if (const char* e = getenv("ICICLE_HIP_NTT_GROUP_MB")) group_mb = (size_t)atoi(e);
      if (group_mb > 0) {
        const size_t row_bytes = (size_t)n * 4;
        rows_per_group = (uint32_t)std::max<size_t>(1, std::min<size_t>(nl.nbatch, (group_mb << 20) / row_bytes));
      }
    }
    const bool grouped = rows_per_group < nl.nbatch;
    // Interleaved transforms (columns_batch, extension field): lane-native tiles, see ntt_fast.hpp. ltot transforms sit
    // word by word at element stride es = ltot; one row group per batch row (row-major) or a single one (columns_batch).
    static const bool lanes_on = !(getenv("ICICLE_HIP_NTT_LANES") && atoi(getenv("ICICLE_HIP_NTT_LANES")) == 0);
    // (A ragged lane count -- 100 columns -- is NOT slow because of its masked last slice: with the 4 surplus lanes split off into a
    //  row-major side batch the three full slices took as long as the four did, 0.85 vs 0.84 ms at 2^20 x 100. Rows of 100 words are
    //  400 bytes apart, so every 128-byte access straddles three 64-byte sectors instead of two: 2.9 TB/s per pass against 4.8 for the
    //  row batch. Built, measured, removed: profiles/r05_notes.md.)
    const uint32_t ltot = cfg->columns_batch ? (uint32_t)batch * lanes : lanes;
    const bool lane_native = lanes_on && fast && ltot > 1;
    const uint32_t row_groups = cfg->columns_batch ? 1u : (uint32_t)batch;
    // Ragged interleaved layouts (columns_batch with a lane count that is not a multiple of 32 words: the Rust suite's 100
    // columns): the WORK buffer pads the lane count to the next multiple of 32, so that only the two passes that touch the
    // caller's buffers see rows that straddle sectors (ntt_plan.h NttLaunch::es_out). ICICLE_HIP_NTT_PAD_LANES=0: off (A/B).
    static const bool pad_on = !(getenv("ICICLE_HIP_NTT_PAD_LANES") && atoi(getenv("ICICLE_HIP_NTT_PAD_LANES")) == 0);
    const bool pad_w = pad_on && lane_native && cfg->columns_batch && P >= 2 && !prerev && !rn_native && !grouped && ltot > 32 && ltot % 32 != 0;
    const uint64_t es_w = pad_w ? (uint64_t)((ltot + 31) / 32) * 32 : nl.es;
    if (P >= 2 && !rn_native) {
      HIP_TRY(d_work.alloc(pad_w ? (size_t)n * es_w * 4 : (grouped ? (size_t)rows_per_group * n * 4 : bytes), st), ICICLE_ALLOCATION_FAILED);
      W = d_work.as<uint32_t>();
    }
    KernelTimer::begin(1, st);
    if (prerev) {
      if (!cfg->columns_batch && lanes == 1 && logn >= 10 && nl.nbatch <= 65535) {
        k_bitrev_rows_tiled<<<dim3((unsigned)(n >> 10), nl.nbatch), 256, 0, st>>>(d_in, W, (uint32_t)logn, nl.bs);
      } else {
        const uint64_t tot = n * nl.nbatch;
        k_bitrev_rows_simple<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(d_in, W, (uint32_t)logn, nl.nbatch, lanes, nl.bs, nl.es, cfg->columns_batch != 0);
      }
      LAUNCH_CHECK("k_bitrev_rows", st);
      nl.in_rev = 0; // the passes below see natural-order rows in W
    }
    for (uint32_t g0 = 0; g0 < nl.nbatch; g0 += rows_per_group) {
    nl.row0 = g0;
    nl.nrows_launch = std::min<uint32_t>(rows_per_group, nl.nbatch - g0);
    for (int p = 0; p < P; p++) {
      const uint32_t* src = rn_native ? (p == 0 ? d_in : d_out) : ((p == 0 && !prerev) ? d_in : W);
      uint32_t* dst = (rn_native || p == P - 1) ? d_out : W; // (RN: every pass is in place once pass 0 has moved the rows to the output)
      nl.src_rel = (grouped && p != 0) ? 1 : 0;
      nl.dst_rel = (grouped && p != P - 1) ? 1 : 0;
      const uint64_t L = (uint64_t)1 << parts[p];
      // fast path: block = T * L/16 threads (<= 512), LDS = 2 buffers of L*(T+1) words (<= 160 KiB)
      const uint64_t epb = L >= 16 ? 16 : L;
      // column passes of 512 / 1024 rows (transforms of 2^25 points and more): 1024-thread blocks, twice the tile width
      static const bool big_on = !(getenv("ICICLE_HIP_NTT_BIG") && atoi(getenv("ICICLE_HIP_NTT_BIG")) == 0);
      const bool cvar_p = nl.coset && (nl.inverse ? p == P - 1 : p == 0);
      // Measured (profiles/r03_notes.md section 9): 2^27 x 4 5.63 -> 5.18 ms, 2^27 x 8 9.74 -> 9.15, but 2^26 x 16 7.25 -> 7.52 and
      // 2^25 x 32 unchanged (one 16-wave block per CU hides less latency than two 8-wave ones): only from 2^27 up.
      const bool big = big_on && fast && !rn_native && !lane_native && logn >= 27 && p < P - 1 && P >= 2 && (parts[p] == 9 || parts[p] == 10) && !cvar_p;
      uint32_t tmax = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(32, (big ? 1024 : 512) * epb / L));
      while (tmax > 1 && 2 * L * (tmax + 1) * 4 > 160 * 1024)
        tmax >>= 1;
      const int rn_mode = !rn_native ? 0 : ((p == 0 && !lane_native) ? 2 : 1);
      while (rn_mode == 2 && tmax > 1 && 2 * (L + (L >> 4)) * tmax * 4 > 160 * 1024)
        tmax >>= 1;
      // lane-native: the tile's tmax word-columns are (tmax >> lsh) logical columns x 2^lsh interleaved transforms. A lane count
      // that is not a multiple of the widest slice keeps a masked last slice (running the remainder as a second launch on
      // narrower slices -- 100 columns = 3 x 32 + a launch of 4-lane tiles -- was built and measured SLOWER, 2^20 x 100:
      // 0.79 -> 0.99 ms: the tail's logical columns are es * 4 bytes apart, 16-byte runs; profiles/r04_notes.md section 1).
      uint32_t lsh = 0;
      if (lane_native)
        while ((2u << lsh) <= tmax && (1u << lsh) < ltot)
          lsh++;
      PassDesc pd = rn_native ? make_pass_rn(parts, P, p, dom.log_max, tmax >> lsh) : make_pass(parts, P, p, n, dom.log_max, tmax >> lsh);
      // Few launch rows (16-64 interleaved transforms = one or two slices): adjacent logical columns that share a twiddle set
      // run as launch rows of one block -- pass 0 (its inter-pass factor depends on column / cprime only) and the last pass
      // (none at all); not the coset / bit-reversed-output variants, whose per-block constants depend on the column.
      uint32_t cg = 1, ag_for_pass = 1;
      const uint32_t tcl = (uint32_t)pd.T; // logical columns in the LDS tile
      {
        static const uint32_t cg_max = getenv("ICICLE_HIP_NTT_COLUMN_GROUP") ? (uint32_t)std::max(1, atoi(getenv("ICICLE_HIP_NTT_COLUMN_GROUP"))) : 8u;
        const bool cvar_here = nl.coset && (nl.inverse ? p == P - 1 : p == 0);
        // (only with full slices: grouping multiplies the mostly idle rows of a ragged last slice as well -- 2^20 x 100: 0.84 -> 0.89 ms)
        const bool full = lane_native && fast && ltot % (1u << lsh) == 0;
        // (RN: every pass can group adjacent logical columns -- the factor behind a pass is rebuilt per row, ntt_fast.hpp rn_rowfac)
        const bool allowed = full && !cvar_here && (rn_native ? P >= 2 : ((p == 0 && P >= 2) || p == P - 1)); // (round 5: the bit-reversed-output store too)
        const bool middle = full && !rn_native && P == 3 && p == 1; // groups over the outer index instead (ntt_plan.h agrp)
        const uint32_t rows_now = row_groups * ((ltot + (1u << lsh) - 1) >> lsh);
        uint32_t want = 1;
        while ((allowed || middle) && want * 2 <= cg_max && rows_now * want * 2 <= 8)
          want *= 2;
        uint32_t ag = 1;
        if (middle) {
          while (want > 1 && ((uint64_t)1 << parts[0]) % want != 0)
            want >>= 1;
          ag = want;
          want = 1;
          pd.ntiles /= ag; // tiles enumerate (a / ag, ct)
          ag_for_pass = ag;
        }
        while (want > 1) {
          const PassDesc pg = rn_native ? make_pass_rn(parts, P, p, dom.log_max, tcl * want) : make_pass(parts, P, p, n, dom.log_max, tcl * want);
          if ((uint32_t)pg.T == tcl * want && (rn_native || pg.is_last || (uint32_t)pg.T <= pg.cprime)) {
            pd = pg;
            cg = want;
            break;
          }
          want >>= 1;
        }
      }
      const uint32_t tw = tcl << lsh; // word-columns per tile
      const uint32_t ag_rows = ag_for_pass;
      static const bool xcd_on = !(getenv("ICICLE_HIP_NTT_XCD") && atoi(getenv("ICICLE_HIP_NTT_XCD")) == 0);
      pd.xcd_remap = (xcd_on && fast && tw < 32 && pd.ntiles >= 64 && pd.ntiles % 8 == 0) ? 1 : 0;
      if (fast) {
        NttLaunch nlp = nl;
        if (pad_w) { // the work buffer's rows are es_w words apart, the caller's ltot
          nlp.es = (src == W) ? es_w : nl.es;
          nlp.es_out = (dst == W) ? es_w : nl.es;
        }
        if (lane_native) { // launch rows = slices of 2^lsh transforms
          nlp.lsh = lsh;
          nlp.ltot = ltot;
          nlp.lanes = (ltot + (1u << lsh) - 1) >> lsh;
          nlp.bs = n * lanes;
          nlp.row0 = 0;
          nlp.tcl = tcl;
          nlp.cgrp = cg * ag_rows;
          nlp.agrp = ag_rows;
          nlp.cst_in = ag_rows > 1 ? pd.in_base_a * nl.es : (uint64_t)tcl * pd.in_st * nl.es;
          nlp.cst_out = (pd.is_last && ag_rows == 1 && !rn_native) ? (uint64_t)tcl * nl.es : nlp.cst_in; // (RN passes are in place: same layout both sides)
          if (pd.is_last && nl.out_rev && !rn_native) nlp.cst_out = 0; // (the bit-reversed store computes the column's place itself, ntt_fast.hpp)
          nlp.nrows_launch = row_groups * nlp.lanes * nlp.cgrp;
        }
        const unsigned threads = (unsigned)(tw * (L / epb));
        const size_t lds_bytes = rn_mode == 2 ? (size_t)2 * tw * (L + (L >> 4)) * 4 : (size_t)2 * L * (tw + 1) * 4;
        nlp.vec4 = (rn_mode == 2 && nl.es == 1 && nl.bs % 4 == 0 && L >= 16 && (((uintptr_t)src) & 15) == 0) ? 1 : 0;
        // rows of the batch handled by one block (twiddles are loaded once per block): aim for >= 1024 blocks = two rounds of
        // the 2 x 256 resident ones (4096 through round 3; same box, profiles/r04_ntt_twiddle_ab.txt: 2^20 x 256 1.39 -> 1.32 ms,
        // 2^16 x 1024 0.272 -> 0.244, 2^24 x 64 5.50 -> 5.45; the per-block twiddle prologue is what the extra rows amortise)
        const uint64_t total_blocks = (uint64_t)pd.ntiles * nlp.nrows_launch;
        static const uint64_t min_blocks = getenv("ICICLE_HIP_NTT_MIN_BLOCKS") ? std::max(1, atoi(getenv("ICICLE_HIP_NTT_MIN_BLOCKS"))) : 1024;
        const uint32_t rpb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nlp.nrows_launch, total_blocks / min_blocks));
        const uint32_t gy = (nlp.nrows_launch + rpb - 1) / rpb;
        // the coset factors touch the first pass (forward) or the last one (inverse) only
        const bool cvar = nl.coset && (nl.inverse ? pd.is_last != 0 : p == 0);
        // 16-byte lanes need unit element stride, rows and bases on 16-byte boundaries and full 32-column tiles
        static const bool v4_on = !(getenv("ICICLE_HIP_NTT_V4") && atoi(getenv("ICICLE_HIP_NTT_V4")) == 0);
        const bool v4 = v4_on && pd.is_last && lanes == 1 && nl.es == 1 && nl.bs % 4 == 0 && pd.T == 32 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
        const bool orev = nl.out_rev != 0 && pd.is_last != 0;
        pass_fn_t fn = rn_mode ? (lane_native ? pick_pass_rn_lanes<PR>(pd.s, cvar) : pick_pass_rn_t<PR, false>(pd.s, rn_mode, cvar))
                       : lane_native ? pick_pass_lanes<PR>(pd.s, pd.is_last != 0, nl.inverse != 0, cvar, orev) : pick_pass<PR>(pd.s, pd.is_last != 0, nl.inverse != 0, cvar, orev, v4, big && threads > 512);
        if (!fn) return ICICLE_INVALID_ARGUMENT;
        HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), ICICLE_INVALID_ARGUMENT);
        fn<<<dim3(pd.ntiles, gy), threads, lds_bytes, st>>>(src, dst, dom.tw, d_ctab.as<uint32_t>(), pd, nlp, rpb);
      }
      if (!fast) {
        const uint32_t tot = (uint32_t)(L * pd.T);
        const unsigned threads = std::max(64u, std::min(1024u, tot / 2));
        HIP_TRY(hipFuncSetAttribute((const void*)k_ntt_pass_generic<PR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), ICICLE_INVALID_ARGUMENT);
        for (uint32_t r0 = 0; r0 < nl.nbatch; r0 += 65535) {
          NttLaunch ns = nl;
          ns.row0 = r0;
          k_ntt_pass_generic<PR><<<dim3(pd.ntiles, std::min<uint32_t>(65535, nl.nbatch - r0)), threads, (size_t)tot * 4, st>>>(src, dst, dom.tw, d_pw.as<uint32_t>(), pd, ns);
        }
      }
      LAUNCH_CHECK("k_ntt_pass", st);
    }
    }
    KernelTimer::end(1, st);
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);

    if (!cfg->are_outputs_on_device) {
      HIP_TRY(hipMemcpyAsync(output, d_out, bytes, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
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

#define DEFINE_NTT_U32(F)                                                                                              \
  extern "C" icicle_error_t F##_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* config, uint32_t* output) \
  {                                                                                                                    \
    GUARDED(ntt_run<F##_params>(input, size, dir, config, output, 1));                                                 \
  }                                                                                                                    \
  extern "C" icicle_error_t F##_extension_ntt(const uint32_t* input, int size, int dir, const icicle_ntt_config_u32_t* config, uint32_t* output) \
  {                                                                                                                    \
    GUARDED(ntt_run<F##_params>(input, size, dir, config, output, 4));                                                 \
  }                                                                                                                    \
  extern "C" icicle_error_t F##_ntt_init_domain(const uint32_t* primitive_root, const icicle_ntt_init_domain_config_t* config) \
  {                                                                                                                    \
    GUARDED(ntt_init_domain_run<F##_params>(primitive_root, config));                                                  \
  }                                                                                                                    \
  extern "C" icicle_error_t F##_ntt_release_domain(void) { GUARDED(ntt_release_domain_run<F##_params>()); }            \
  extern "C" icicle_error_t F##_get_root_of_unity(uint64_t max_size, uint32_t* rou)                                    \
  {                                                                                                                    \
    GUARDED(get_root_of_unity_run<F##_params>(max_size, rou));                                                         \
  }                                                                                                                    \
  extern "C" icicle_error_t F##_get_root_of_unity_from_domain(uint64_t logn, uint32_t* rou)                            \
  {                                                                                                                    \
    GUARDED(ntt_rou_from_domain_run<F##_params>(logn, rou));                                                           \
  }

// collision-free aliases for the reference-runtime plugin (see msm.hip)
#define DEFINE_NTT_HELPERS(F)                                                                                          \
  extern "C" icicle_error_t F##_hip_twiddle_rows(uint32_t* data, uint64_t rows, uint64_t cols, uint64_t row0, uint32_t logn_total, bool inverse, icicleStreamHandle stream) \
  {                                                                                                                    \
    GUARDED(twiddle_rows_run<F##_params>(data, rows, cols, row0, logn_total, inverse, (hipStream_t)stream));           \
  }
DEFINE_NTT_HELPERS(babybear)
DEFINE_NTT_HELPERS(koalabear)

#define DEFINE_NTT_ALIASES(F)                                                                                          \
  extern "C" icicle_error_t icicle_hip_##F##_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u32_t* c, uint32_t* o) { GUARDED(ntt_run<F##_params>(i, n, d, c, o, 1)); } \
  extern "C" icicle_error_t icicle_hip_##F##_extension_ntt(const uint32_t* i, int n, int d, const icicle_ntt_config_u32_t* c, uint32_t* o) { GUARDED(ntt_run<F##_params>(i, n, d, c, o, 4)); } \
  extern "C" icicle_error_t icicle_hip_##F##_ntt_init_domain(const uint32_t* r, const icicle_ntt_init_domain_config_t* c) { GUARDED(ntt_init_domain_run<F##_params>(r, c)); } \
  extern "C" icicle_error_t icicle_hip_##F##_ntt_release_domain(void) { GUARDED(ntt_release_domain_run<F##_params>()); } \
  extern "C" icicle_error_t icicle_hip_##F##_get_root_of_unity_from_domain(uint64_t l, uint32_t* r) { GUARDED(ntt_rou_from_domain_run<F##_params>(l, r)); }
DEFINE_NTT_ALIASES(babybear)
DEFINE_NTT_ALIASES(koalabear)

DEFINE_NTT_U32(babybear)
DEFINE_NTT_U32(koalabear)
