// MSM for gfx950: sort-based Pippenger bucket method.
//
// Reference semantics: icicle/backend/cpu/src/curve/cpu_msm.hpp:431-443 (batch / shared-bases /
// precompute layout), :259-314 (signed digits, skip zero bases), :455-480 (precompute contract).
// The CPU backend gives each worker private buckets and random-access RMWs into them; on MI355X
// the same sum is reorganised so that the only random access left is a read-only gather:
//
//   1. bases -> packed Montgomery copy in HBM (one pass, 2 field muls per point)
//   2. scalars -> signed c-bit digits, one u32 per (window, scalar), coalesced writes
//   3. per window: counting sort of point indices by bucket. Histogram and cursors live in LDS
//      (2^(c-1) u32 counters = 128 KiB at c = 16, of gfx950's 160 KiB/CU); ds_add_rtn runs at
//      ~550 G atomics/s chip-wide vs ~27 G/s for global atomics (profiles/r01_alu_ubench.txt).
//   4. bucket accumulation: ONE THREAD PER BUCKET walks its sorted index list, gathers the 64 B
//      affine point and does an XYZZ mixed add entirely in registers -- >95 % of all time;
//      integer-ALU bound (v_mad_u64_u32), see DESIGN.md.
//   5. bucket reduction: per-segment running sums (complete projective adds), wave-level tree,
//      Horner over windows.
//
// Everything is enqueued on config.stream with stream-ordered temporaries; the host only blocks
// when the API contract requires it (is_async == false or results on host).
#include "common.h"
#include "ec.cuh"
#include <algorithm>

namespace icicle_hip {

  struct MsmPlan {
    int bits;    // scalar bits considered
    int c;       // window bits
    int nwin;    // total windows W = ceil((bits+1)/c)
    int pf;      // precompute factor
    int wpf;     // windows per precomputed base = target windows actually accumulated
    uint32_t nb; // buckets per window = 2^(c-1)
  };

  static MsmPlan make_plan(int n, int scalar_bits, const icicle_msm_config_t& cfg)
  {
    MsmPlan p;
    p.bits = (cfg.bitsize > 0 && cfg.bitsize < scalar_bits) ? cfg.bitsize : scalar_bits;
    p.pf = std::max(1, cfg.precompute_factor);
    int c = cfg.c;
    if (c <= 0) {
      // minimise  (#mixed adds) + (bucket-reduction adds, weighted for their poor parallelism)
      double best = 1e300;
      for (int cc = 2; cc <= 16; cc++) {
        const int w = (p.bits + 1 + cc - 1) / cc;
        const int wpf = (w + p.pf - 1) / p.pf;
        const double cost = (double)w * n + 8.0 * wpf * (double)(1u << (cc - 1));
        if (cost < best) {
          best = cost;
          c = cc;
        }
      }
    }
    c = std::min(16, std::max(2, c)); // LDS histogram holds 2^(c-1) counters
    p.c = c;
    p.nwin = (p.bits + 1 + c - 1) / c;
    p.wpf = (p.nwin + p.pf - 1) / p.pf;
    p.nb = 1u << (c - 1);
    return p;
  }

  // ------------------------------------------------------------------------------------------
  // 1. bases -> packed Montgomery (thread per coordinate)
  template <class C>
  __global__ __launch_bounds__(256) void k_bases_to_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t ncoord, bool in_refmont)
  {
    using F = typename EC<C>::F;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= ncoord) return;
    uint32_t w[F::N32];
#pragma unroll
    for (int i = 0; i < F::N32; i++)
      w[i] = in[t * F::N32 + i];
    typename F::fe v = in_refmont ? F::from_refmont(w) : F::from_canonical(w);
    F::pack(w, F::reduce(v));
#pragma unroll
    for (int i = 0; i < F::N32; i++)
      out[t * F::N32 + i] = w[i];
  }

  // ------------------------------------------------------------------------------------------
  // 2. signed-digit decomposition. digit word = |d| | (d<0)<<31, |d| in [0, 2^(c-1)], 0 = skip.
  template <class C>
  __global__ __launch_bounds__(256) void k_digits(const uint32_t* __restrict__ scalars, uint32_t* __restrict__ dig, int n, int c, int nwin, bool scalars_refmont)
  {
    using FR = FieldOps<typename C::fr>;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[FR::N32 + 1];
#pragma unroll
    for (int k = 0; k < FR::N32; k++)
      w[k] = scalars[(size_t)i * FR::N32 + k];
    if (scalars_refmont) { // x*2^(32*N32) -> x  (cpu_msm.hpp:274-275 from_montgomery)
      typename FR::fe cst;
#pragma unroll
      for (int k = 0; k < FR::N; k++)
        cst.l[k] = C::fr::REFMONT_TO_CANON[k];
      BF_SET_BOUND(cst, 1);
      FR::pack(w, FR::reduce(FR::mul(FR::unpack(w), cst)));
    }
    w[FR::N32] = 0;
    const uint32_t half = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    uint32_t carry = 0;
    for (int wi = 0; wi < nwin; wi++) {
      const int bit = wi * c;
      const int word = bit >> 5, sh = bit & 31;
      uint32_t v = 0;
      if (word < FR::N32) {
        uint64_t two = ((uint64_t)w[word + 1] << 32) | w[word];
        v = (uint32_t)(two >> sh) & mask;
      }
      v += carry;
      uint32_t d, neg;
      if (v > half) {
        d = (1u << c) - v;
        neg = 1;
        carry = 1;
      } else {
        d = v;
        neg = 0;
        carry = 0;
      }
      dig[(size_t)wi * n + i] = d | (d ? (neg << 31) : 0);
    }
  }

  // ------------------------------------------------------------------------------------------
  // 3. counting sort per target window, LDS-privatised histogram / cursors.
  //    grid = (B, wpf); block b owns scalars [b*chunk, (b+1)*chunk).
  __global__ __launch_bounds__(1024) void k_hist(const uint32_t* __restrict__ dig, uint32_t* __restrict__ blockhist, int n, int chunk, int nwin, int wpf, int pf, uint32_t nb)
  {
    extern __shared__ uint32_t lds[];
    const int b = blockIdx.x, wp = blockIdx.y, B = gridDim.x;
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x)
      lds[k] = 0;
    __syncthreads();
    const int lo = b * chunk, hi = min(n, lo + chunk);
    for (int j = 0; j < pf; j++) {
      const int w = j * wpf + wp;
      if (w >= nwin) break;
      const uint32_t* d = dig + (size_t)w * n;
      for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t k = d[i] & 0x7fffffffu;
        if (k) atomicAdd(&lds[k - 1], 1u);
      }
    }
    __syncthreads();
    uint32_t* out = blockhist + ((size_t)wp * B + b) * nb;
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x)
      out[k] = lds[k];
  }

  // exclusive prefix over blocks for every (window,bucket); total -> count
  __global__ __launch_bounds__(256) void k_scan_blocks(uint32_t* __restrict__ blockhist, uint32_t* __restrict__ count, int B, uint32_t nb, int wpf)
  {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= (size_t)wpf * nb) return;
    const size_t wp = t / nb, k = t % nb;
    uint32_t run = 0;
    for (int b = 0; b < B; b++) {
      uint32_t* p = blockhist + (wp * B + b) * nb + k;
      const uint32_t x = *p;
      *p = run;
      run += x;
    }
    count[t] = run;
  }

  // exclusive scan over the buckets of one window: one 1024-thread block per window
  __global__ __launch_bounds__(1024) void k_scan_buckets(const uint32_t* __restrict__ count, uint32_t* __restrict__ offs, uint32_t nb)
  {
    __shared__ uint32_t part[1024];
    const int wp = blockIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(nb, lo + per);
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; k++)
      s += count[(size_t)wp * nb + k];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int d = 1; d < 1024; d <<= 1) {
      uint32_t v = (threadIdx.x >= (unsigned)d) ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t k = lo; k < hi; k++) {
      offs[(size_t)wp * nb + k] = run;
      run += count[(size_t)wp * nb + k];
    }
  }

  __global__ __launch_bounds__(1024) void k_scatter(const uint32_t* __restrict__ dig, const uint32_t* __restrict__ blockhist, const uint32_t* __restrict__ offs, uint32_t* __restrict__ sorted, int n, int chunk, int nwin, int wpf, int pf, uint32_t nb, size_t cap)
  {
    extern __shared__ uint32_t lds[];
    const int b = blockIdx.x, wp = blockIdx.y, B = gridDim.x;
    const uint32_t* bh = blockhist + ((size_t)wp * B + b) * nb;
    const uint32_t* of = offs + (size_t)wp * nb;
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x)
      lds[k] = of[k] + bh[k];
    __syncthreads();
    uint32_t* dst = sorted + (size_t)wp * cap;
    const int lo = b * chunk, hi = min(n, lo + chunk);
    for (int j = 0; j < pf; j++) {
      const int w = j * wpf + wp;
      if (w >= nwin) break;
      const uint32_t* d = dig + (size_t)w * n;
      for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t e = d[i];
        const uint32_t k = e & 0x7fffffffu;
        if (k) {
          const uint32_t pos = atomicAdd(&lds[k - 1], 1u);
          dst[pos] = ((uint32_t)i * (uint32_t)pf + (uint32_t)j) | (e & 0x80000000u);
        }
      }
    }
  }

  // ------------------------------------------------------------------------------------------
  // 4. bucket accumulation: thread per (window,bucket), XYZZ accumulator in registers.
  template <class C>
  __global__ __launch_bounds__(128) void k_accumulate(const uint32_t* __restrict__ bases_mont, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ count, const uint32_t* __restrict__ offs, typename EC<C>::Proj* __restrict__ buckets, uint32_t nb, int wpf, size_t cap)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32; // words per affine point
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= (size_t)wpf * nb) return;
    const size_t wp = t / nb;
    const uint32_t cnt = count[t];
    const uint32_t* src = sorted + wp * cap + offs[t];
    typename E::XYZZ acc;
    bool empty = true;
    for (uint32_t j = 0; j < cnt; j++) {
      const uint32_t e = src[j];
      const uint4* p = reinterpret_cast<const uint4*>(bases_mont + (size_t)(e & 0x7fffffffu) * PW);
      uint32_t w[PW];
#pragma unroll
      for (int q = 0; q < PW / 4; q++) {
        const uint4 v = p[q];
        w[4 * q] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
      }
      if (E::words_are_zero(w)) continue; // identity base: contributes nothing (cpu_msm.hpp:282)
      typename E::Aff a = E::cneg(E::load_mont(w), (e >> 31) != 0);
      E::madd(acc, empty, a);
    }
    buckets[t] = E::to_proj(acc, empty);
  }

  // ------------------------------------------------------------------------------------------
  // 5a. per-segment running sums. Segment = m consecutive buckets [k0, k0+m) of one window;
  //     val = sum_{k} (k+1) * B_k  =  tri + k0 * line   (bucket index k carries weight k+1).
  template <class C>
  __global__ __launch_bounds__(64) void k_reduce_segments(const typename EC<C>::Proj* __restrict__ buckets, typename EC<C>::Proj* __restrict__ segval, uint32_t nb, uint32_t m, int wpf)
  {
    using E = EC<C>;
    const uint32_t nseg = nb / m;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= (size_t)wpf * nseg) return;
    const size_t wp = t / nseg;
    const uint32_t seg = t % nseg;
    const uint32_t k0 = seg * m;
    const typename E::Proj* b = buckets + wp * nb + k0;
    typename E::Proj line = E::proj_identity(), tri = E::proj_identity();
    for (int k = (int)m - 1; k >= 0; k--) {
      line = E::add(line, b[k]);
      tri = E::add(tri, line);
    }
    if (k0) tri = E::add(tri, E::mul_small(line, k0));
    segval[t] = tri;
  }

  // 5b. one wave per window: each lane folds nseg/64 segment values, then a 64-wide tree through LDS
  template <class C>
  __global__ __launch_bounds__(64) void k_reduce_window(const typename EC<C>::Proj* __restrict__ segval, typename EC<C>::Proj* __restrict__ winsum, uint32_t nseg)
  {
    using E = EC<C>;
    __shared__ typename E::Proj sh[64];
    const int wp = blockIdx.x, lane = threadIdx.x;
    typename E::Proj v = E::proj_identity();
    for (uint32_t s = lane; s < nseg; s += 64)
      v = E::add(v, segval[(size_t)wp * nseg + s]);
    sh[lane] = v;
    __syncthreads();
    for (int s = 32; s >= 1; s >>= 1) {
      if (lane < s) {
        v = E::add(v, sh[lane + s]);
        sh[lane] = v;
      }
      __syncthreads();
    }
    if (lane == 0) winsum[wp] = v;
  }

  // 5c. window combine: result = sum_w 2^(c*w) * winsum[w], written in the reference's
  //     projective_t layout (canonical words). One 128-lane block: lane w scales its own window
  //     sum by c*w doublings (the same critical path as a serial Horner, but the doublings of
  //     different windows overlap), then a tree through LDS. <1 % of the work.
  //     (A <<<1,1>>> serial Horner is provably wave-uniform, so hipcc compiles ALL of its field
  //     arithmetic to SALU code; that variant returned wrong results on gfx950 for some shapes.)
  template <class C>
  __global__ __launch_bounds__(128) void k_final(const typename EC<C>::Proj* __restrict__ winsum, uint32_t* __restrict__ result, int wpf, int c)
  {
    using E = EC<C>;
    __shared__ typename E::Proj sh[128];
    const int lane = threadIdx.x;
    typename E::Proj v = E::proj_identity();
    if (lane < wpf) {
      v = winsum[lane];
      for (int i = 0; i < lane * c; i++)
        v = E::dbl(v);
    }
    sh[lane] = v;
    __syncthreads();
    for (int s = 64; s >= 1; s >>= 1) {
      if (lane < s) {
        v = E::add(v, sh[lane + s]);
        sh[lane] = v;
      }
      __syncthreads();
    }
    if (lane == 0) E::store_proj_canonical(result, v);
  }

  // ------------------------------------------------------------------------------------------
  // field inversion a^(p-2) (precompute / generator paths only)
  template <class C>
  __device__ typename EC<C>::fe fe_inv(const typename EC<C>::fe& a)
  {
    using F = typename EC<C>::F;
    typename F::fe r = F::one(), base = a;
    for (int wi = 0; wi < F::N32; wi++) {
      uint32_t e = C::fq::P32[wi] - (wi == 0 ? 2u : 0u);
      for (int b = 0; b < 32; b++) {
        if ((e >> b) & 1) r = F::mul(r, base);
        base = F::sqr(base);
      }
    }
    return r;
  }

  // projective (Montgomery) -> affine words; identity -> (0,0)
  template <class C>
  __device__ void store_affine(uint32_t* w, const typename EC<C>::Proj& p, bool refmont)
  {
    using E = EC<C>;
    using F = typename E::F;
    if (F::is_zero(p.z)) {
      for (int i = 0; i < 2 * E::N32; i++)
        w[i] = 0;
      return;
    }
    typename F::fe zi = fe_inv<C>(p.z);
    typename F::fe x = F::mul(p.x, zi), y = F::mul(p.y, zi);
    if (refmont) {
      F::to_refmont(w, x);
      F::to_refmont(w + E::N32, y);
    } else {
      F::to_canonical(w, x);
      F::to_canonical(w + E::N32, y);
    }
  }

  // msm_precompute_bases: out[pf*i + j] = 2^(j*shift) * P_i  (cpu_msm.hpp:455-480), shift = c*wpf
  template <class C>
  __global__ __launch_bounds__(64) void k_precompute(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n, int pf, int shift, bool refmont)
  {
    using E = EC<C>;
    using F = typename E::F;
    constexpr int PW = 2 * E::N32;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[PW];
    for (int k = 0; k < PW; k++) {
      w[k] = in[(size_t)i * PW + k];
      out[(size_t)i * pf * PW + k] = w[k];
    }
    typename E::Proj p;
    if (E::words_are_zero(w)) {
      p = E::proj_identity();
    } else {
      typename E::Aff a;
      a.x = refmont ? F::from_refmont(w) : F::from_canonical(w);
      a.y = refmont ? F::from_refmont(w + E::N32) : F::from_canonical(w + E::N32);
      p = E::to_proj(a);
    }
    for (int j = 1; j < pf; j++) {
      for (int s = 0; s < shift; s++)
        p = E::dbl(p);
      uint32_t o[PW];
      store_affine<C>(o, p, refmont);
      for (int k = 0; k < PW; k++)
        out[((size_t)i * pf + j) * PW + k] = o[k];
    }
  }

  // synthetic distinct points (k0 + i) * G, i < n; each thread produces L consecutive points
  template <class C>
  __global__ __launch_bounds__(64) void k_generate(uint32_t* __restrict__ out, int n, uint64_t k0, int L)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const long long first = (long long)t * L;
    if (first >= n) return;
    const typename E::Proj g = E::to_proj(E::generator());
    uint64_t k = k0 + (uint64_t)first;
    typename E::Proj p = E::proj_identity();
    for (int b = 63; b >= 0; b--) {
      p = E::dbl(p);
      if ((k >> b) & 1) p = E::add(p, g);
    }
    for (int j = 0; j < L && first + j < n; j++) {
      uint32_t o[PW];
      store_affine<C>(o, p, false);
      for (int q = 0; q < PW; q++)
        out[((size_t)first + j) * PW + q] = o[q];
      p = E::add(p, g);
    }
  }

  // sum of n projective points in the reference's canonical layout (multi-GPU partial-result combine)
  template <class C>
  __global__ __launch_bounds__(64) void k_proj_sum(const uint32_t* __restrict__ pts, int n, uint32_t* __restrict__ out)
  {
    using E = EC<C>;
    using F = typename E::F;
    __shared__ typename E::Proj sh[64];
    const int lane = threadIdx.x;
    typename E::Proj v = E::proj_identity();
    for (int i = lane; i < n; i += 64) {
      const uint32_t* w = pts + (size_t)i * 3 * E::N32;
      typename E::Proj p;
      p.x = F::from_canonical(w);
      p.y = F::from_canonical(w + E::N32);
      p.z = F::from_canonical(w + 2 * E::N32);
      v = E::add(v, p);
    }
    sh[lane] = v;
    __syncthreads();
    for (int s = 32; s >= 1; s >>= 1) {
      if (lane < s) {
        v = E::add(v, sh[lane + s]);
        sh[lane] = v;
      }
      __syncthreads();
    }
    if (lane == 0) E::store_proj_canonical(out, v);
  }

  template <class C>
  static icicle_error_t proj_sum_run(const void* pts, int n, void* out, hipStream_t st)
  {
    if (n < 0 || !out || (n > 0 && !pts)) return ICICLE_INVALID_ARGUMENT;
    ICICLE_TRY(bind_current_device());
    k_proj_sum<C><<<1, 64, 0, st>>>((const uint32_t*)pts, n, (uint32_t*)out);
    LAUNCH_CHECK("k_proj_sum", st);
    return ICICLE_SUCCESS;
  }

  // ------------------------------------------------------------------------------------------
  template <class C>
  static icicle_error_t msm_run(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v)
  {
    using E = EC<C>;
    using FR = FieldOps<typename C::fr>;
    constexpr int PW = 2 * E::N32, RW = 3 * E::N32;
    if (!cfg || !results_v || n < 0) return ICICLE_INVALID_ARGUMENT;
    const int batch = std::max(1, cfg->batch_size);
    if (n > 0 && (!scalars_v || !bases_v)) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const MsmPlan pl = make_plan(std::max(n, 1), C::fr::NBITS, *cfg);
    const int pf = pl.pf;
    if ((long long)n * pf >= (1ll << 31)) return ICICLE_INVALID_ARGUMENT;
    const bool shared = cfg->are_points_shared_in_batch || batch == 1;
    const size_t npts_one = (size_t)n * pf;
    const size_t npts_all = shared ? npts_one : npts_one * batch;

    // ---- result buffer
    TempBuf d_res_tmp;
    uint32_t* d_res = (uint32_t*)results_v;
    if (!cfg->are_results_on_device) {
      HIP_TRY(d_res_tmp.alloc((size_t)batch * RW * 4, st), ICICLE_ALLOCATION_FAILED);
      d_res = d_res_tmp.as<uint32_t>();
    }

    if (n == 0) { // empty sum = identity for every batch element
      std::vector<uint32_t> id((size_t)batch * RW, 0);
      for (int b = 0; b < batch; b++)
        id[(size_t)b * RW + E::N32] = 1; // (0:1:0)
      HIP_TRY(hipMemcpyAsync(d_res, id.data(), id.size() * 4, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
      if (!cfg->are_results_on_device) HIP_TRY(hipMemcpy(results_v, d_res, id.size() * 4, hipMemcpyDeviceToHost), ICICLE_COPY_FAILED);
      return ICICLE_SUCCESS;
    }

    // ---- stage inputs
    TempBuf d_sc_tmp, d_b_tmp;
    const uint32_t* d_scalars = (const uint32_t*)scalars_v;
    if (!cfg->are_scalars_on_device) {
      const size_t bytes = (size_t)batch * n * FR::N32 * 4;
      HIP_TRY(d_sc_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_sc_tmp.ptr(), scalars_v, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_scalars = d_sc_tmp.as<uint32_t>();
    }
    const uint32_t* d_bases = (const uint32_t*)bases_v;
    if (!cfg->are_points_on_device) {
      const size_t bytes = npts_all * PW * 4;
      HIP_TRY(d_b_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_b_tmp.ptr(), bases_v, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_bases = d_b_tmp.as<uint32_t>();
    }

    // ---- temporaries
    const uint32_t nb = pl.nb;
    const int wpf = pl.wpf;
    const size_t cap = npts_one; // sorted-index capacity per target window
    const int B = (int)std::min<size_t>(64, ((size_t)n + 16383) / 16384);
    const int chunk = (n + B - 1) / B;
    const uint32_t m = std::min<uint32_t>(nb, 32);
    const uint32_t nseg = nb / m;
    TempBuf d_mont, d_dig, d_sorted, d_bh, d_count, d_offs, d_buckets, d_seg, d_win;
    HIP_TRY(d_mont.alloc(npts_one * PW * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_dig.alloc((size_t)pl.nwin * n * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_sorted.alloc((size_t)wpf * cap * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_bh.alloc((size_t)wpf * B * nb * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_count.alloc((size_t)wpf * nb * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_offs.alloc((size_t)wpf * nb * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_buckets.alloc((size_t)wpf * nb * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_seg.alloc((size_t)wpf * nseg * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_win.alloc((size_t)wpf * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);

    const size_t lds_bytes = (size_t)nb * 4;
    HIP_TRY(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 131072), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 131072), ICICLE_INVALID_ARGUMENT);

    for (int b = 0; b < batch; b++) {
      if (b == 0 || !shared) {
        const uint32_t* src = d_bases + (shared ? 0 : (size_t)b * npts_one * PW);
        const size_t ncoord = npts_one * 2;
        k_bases_to_mont<C><<<dim3((unsigned)((ncoord + 255) / 256)), 256, 0, st>>>(src, d_mont.as<uint32_t>(), ncoord, cfg->are_points_montgomery_form);
        LAUNCH_CHECK("k_bases_to_mont", st);
      }
      k_digits<C><<<(n + 255) / 256, 256, 0, st>>>(d_scalars + (size_t)b * n * FR::N32, d_dig.as<uint32_t>(), n, pl.c, pl.nwin, cfg->are_scalars_montgomery_form);
      LAUNCH_CHECK("k_digits", st);
      k_hist<<<dim3(B, wpf), 1024, lds_bytes, st>>>(d_dig.as<uint32_t>(), d_bh.as<uint32_t>(), n, chunk, pl.nwin, wpf, pf, nb);
      LAUNCH_CHECK("k_hist", st);
      const size_t nbk = (size_t)wpf * nb;
      k_scan_blocks<<<(unsigned)((nbk + 255) / 256), 256, 0, st>>>(d_bh.as<uint32_t>(), d_count.as<uint32_t>(), B, nb, wpf);
      LAUNCH_CHECK("k_scan_blocks", st);
      k_scan_buckets<<<wpf, 1024, 0, st>>>(d_count.as<uint32_t>(), d_offs.as<uint32_t>(), nb);
      LAUNCH_CHECK("k_scan_buckets", st);
      k_scatter<<<dim3(B, wpf), 1024, lds_bytes, st>>>(d_dig.as<uint32_t>(), d_bh.as<uint32_t>(), d_offs.as<uint32_t>(), d_sorted.as<uint32_t>(), n, chunk, pl.nwin, wpf, pf, nb, cap);
      LAUNCH_CHECK("k_scatter", st);
      KernelTimer::begin(0, st);
      k_accumulate<C><<<(unsigned)((nbk + 127) / 128), 128, 0, st>>>(d_mont.as<uint32_t>(), d_sorted.as<uint32_t>(), d_count.as<uint32_t>(), d_offs.as<uint32_t>(), d_buckets.as<typename E::Proj>(), nb, wpf, cap);
      LAUNCH_CHECK("k_accumulate", st);
      KernelTimer::end(0, st);
      const size_t nsg = (size_t)wpf * nseg;
      k_reduce_segments<C><<<(unsigned)((nsg + 63) / 64), 64, 0, st>>>(d_buckets.as<typename E::Proj>(), d_seg.as<typename E::Proj>(), nb, m, wpf);
      LAUNCH_CHECK("k_reduce_segments", st);
      k_reduce_window<C><<<wpf, 64, 0, st>>>(d_seg.as<typename E::Proj>(), d_win.as<typename E::Proj>(), nseg);
      LAUNCH_CHECK("k_reduce_window", st);
      k_final<C><<<1, 128, 0, st>>>(d_win.as<typename E::Proj>(), d_res + (size_t)b * RW, wpf, pl.c);
      LAUNCH_CHECK("k_final", st);
    }
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);

    if (!cfg->are_results_on_device) {
      HIP_TRY(hipMemcpyAsync(results_v, d_res, (size_t)batch * RW * 4, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    } else if (!cfg->is_async) {
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    }
    return ICICLE_SUCCESS;
  }

  template <class C>
  static icicle_error_t msm_precompute_run(const void* in_v, int n, const icicle_msm_config_t* cfg, void* out_v)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32;
    if (!cfg || n < 0) return ICICLE_INVALID_ARGUMENT;
    if (n == 0) return ICICLE_SUCCESS;
    if (!in_v || !out_v) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const MsmPlan pl = make_plan(n, C::fr::NBITS, *cfg);
    const int pf = pl.pf;
    TempBuf d_in_tmp, d_out_tmp;
    const uint32_t* d_in = (const uint32_t*)in_v;
    uint32_t* d_out = (uint32_t*)out_v;
    // input location: are_points_on_device ; output location: are_results_on_device (msm.h:39-47)
    if (!cfg->are_points_on_device) {
      HIP_TRY(d_in_tmp.alloc((size_t)n * PW * 4, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_in_tmp.ptr(), in_v, (size_t)n * PW * 4, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_in = d_in_tmp.as<uint32_t>();
    }
    if (!cfg->are_results_on_device) {
      HIP_TRY(d_out_tmp.alloc((size_t)n * pf * PW * 4, st), ICICLE_ALLOCATION_FAILED);
      d_out = d_out_tmp.as<uint32_t>();
    }
    k_precompute<C><<<(n + 63) / 64, 64, 0, st>>>(d_in, d_out, n, pf, pl.c * pl.wpf, cfg->are_points_montgomery_form);
    LAUNCH_CHECK("k_precompute", st);
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);
    if (!cfg->are_results_on_device) {
      HIP_TRY(hipMemcpyAsync(out_v, d_out, (size_t)n * pf * PW * 4, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    } else if (!cfg->is_async) {
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    }
    return ICICLE_SUCCESS;
  }

  template <class C>
  static icicle_error_t generate_run(void* out_v, int n, uint64_t k0, bool on_device, hipStream_t st)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32;
    if (n < 0 || (n > 0 && !out_v)) return ICICLE_INVALID_ARGUMENT;
    if (n == 0) return ICICLE_SUCCESS;
    ICICLE_TRY(bind_current_device());
    TempBuf tmp;
    uint32_t* d_out = (uint32_t*)out_v;
    if (!on_device) {
      HIP_TRY(tmp.alloc((size_t)n * PW * 4, st), ICICLE_ALLOCATION_FAILED);
      d_out = tmp.as<uint32_t>();
    }
    const int L = 16;
    const int nthreads = (n + L - 1) / L;
    k_generate<C><<<(nthreads + 63) / 64, 64, 0, st>>>(d_out, n, k0, L);
    LAUNCH_CHECK("k_generate", st);
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);
    if (!on_device) HIP_TRY(hipMemcpyAsync(out_v, d_out, (size_t)n * PW * 4, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
    HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    return ICICLE_SUCCESS;
  }

} // namespace icicle_hip

using namespace icicle_hip;

// exceptions must never cross the C boundary (Rust/Go callers are extern "C" frames)
#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

extern "C" {
icicle_error_t bn254_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results)
{
  GUARDED(msm_run<bn254_g1>(scalars, bases, msm_size, config, results));
}
icicle_error_t bn254_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases)
{
  GUARDED(msm_precompute_run<bn254_g1>(input_bases, nof_bases, config, output_bases));
}
icicle_error_t bls12_381_msm(const void* scalars, const void* bases, int msm_size, const icicle_msm_config_t* config, void* results)
{
  GUARDED(msm_run<bls12_381_g1>(scalars, bases, msm_size, config, results));
}
icicle_error_t bls12_381_msm_precompute_bases(const void* input_bases, int nof_bases, const icicle_msm_config_t* config, void* output_bases)
{
  GUARDED(msm_precompute_run<bls12_381_g1>(input_bases, nof_bases, config, output_bases));
}
icicle_error_t bn254_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream)
{
  GUARDED(proj_sum_run<bn254_g1>(points, n, out, (hipStream_t)stream));
}
icicle_error_t bls12_381_hip_projective_sum(const void* points, int n, void* out, icicleStreamHandle stream)
{
  GUARDED(proj_sum_run<bls12_381_g1>(points, n, out, (hipStream_t)stream));
}
icicle_error_t bn254_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream)
{
  GUARDED(generate_run<bn254_g1>(out, n, k0, out_on_device, (hipStream_t)stream));
}
icicle_error_t bls12_381_hip_generate_affine_points(void* out, int n, uint64_t k0, bool out_on_device, icicleStreamHandle stream)
{
  GUARDED(generate_run<bls12_381_g1>(out, n, k0, out_on_device, (hipStream_t)stream));
}
}
