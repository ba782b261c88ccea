// MSM for gfx950: sort-based Pippenger bucket method.
//
// Reference semantics: icicle/backend/cpu/src/curve/cpu_msm.hpp:431-443 (batch / shared-bases /
// precompute layout), :259-314 (signed digits, skip zero bases), :455-480 (precompute contract).
// The CPU backend gives each worker private buckets and random-access RMWs into them; on MI355X
// the same sum is reorganised so that the only random access left is a read-only gather:
//
//   1. (no Montgomery copy of the bases: k_accumulate gathers the caller's canonical words, see ec.hpp)
//   2. k_digits          scalars -> signed c-bit digits, one u32 per (window, scalar), coalesced
//   3. two-level counting sort of point indices by bucket, every scatter staged through an LDS
//      tile-sort so that HBM sees runs, not 4-byte random writes:
//        pass A  k_a_count / k_scan_a / k_a_scatter     partition by the high hb bits of the key
//        pass B  k_b_plan / k_b_count / k_scan_buckets / k_b_scatter   sort by the low lb bits
//      (c <= 11: pass A alone sorts by the whole key, k_a_scatter<true> + k_tables_from_a)
//   4. k_accumulate      ONE THREAD PER BUCKET walks its sorted index list, gathers the 64 B affine
//      point and does an XYZZ mixed add entirely in registers -- ~77 % of the time at 2^26,
//      integer-ALU bound (v_mad_u64_u32), see DESIGN.md. Buckets longer than `seg` points are
//      split (k_plan_overflow) and their partial sums folded back in parallel (k_fold_overflow).
//   5. bucket reduction: k_reduce_wave (wave64 suffix scans over 64*m-bucket chunks) -> k_reduce_window (256
//      lanes per window) -> k_final (4 lanes per window, Horner over windows).
// A batch of MSMs is folded into the window dimension: all of the above is launched once for up
// to BB MSMs x wpf windows.
//
// Everything is enqueued on config.stream with arena-leased temporaries (common.h TempBuf); the
// host only blocks when the API contract requires it (is_async == false or results on host).
#pragma once
#include "common.h"
#include "ec.hpp"
#include "ec_dbl_quad.hpp"
#include "msm_plan.h"
#include <algorithm>
#include <tuple>

namespace icicle_hip {

  // ------------------------------------------------------------------------------------------
  // 1. bases staging (thread per coordinate) -- NOT on the default path: bucket accumulation gathers the caller's
  //    canonical affine words directly (ec.hpp header). Only bases given in the reference's Montgomery form
  //    (are_points_montgomery_form: x*2^(32*N32) -> x) or at an address that is not 16-byte aligned are copied.
  template <class C>
  __global__ __launch_bounds__(256) void k_bases_stage(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t ncoord, bool in_refmont)
  {
    using BF = FieldOps<typename C::fq>; // per base-field coordinate (G2: 4 per point)
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= ncoord) return;
    uint32_t w[BF::N32];
#pragma unroll
    for (int i = 0; i < BF::N32; i++)
      w[i] = in[t * BF::N32 + i];
    if (in_refmont) {
      typename BF::fe cst;
#pragma unroll
      for (int k = 0; k < BF::N; k++)
        cst.l[k] = C::fq::REFMONT_TO_CANON[k];
      BF_SET_BOUND(cst, 1);
      BF::pack(w, BF::reduce(BF::mul(BF::unpack(w), cst)));
    }
#pragma unroll
    for (int i = 0; i < BF::N32; i++)
      out[t * BF::N32 + i] = w[i];
  }

  // ------------------------------------------------------------------------------------------
  // 2+3. signed digits and a two-level counting sort of point indices by bucket.
  //
  // A single-level scatter into 2^(c-1) bucket lists writes 4 bytes to a random one of 32768 open
  // cache lines per element -- measured 29 ms of the first version's 144 ms (profiles/
  // r01_v1_kernel_stats.txt), almost all write amplification (8.6x, profiles/r01_notes.md). Here
  // every block first sorts its tile in LDS and then writes each bin's run contiguously:
  //   k_digits: one pass over the scalars, digit array dig[window][scalar] in HBM (4 B each).
  //   pass A:   block (b, window) owns digits [b*chunk, (b+1)*chunk) of that window and partitions
  //             them by the HIGH hb bits of the bucket key. Element = sign | low key bits | j |
  //             index within chunk.
  //   pass B:   blocks own sub-chunks (<= 2^17 elements) of one (window, partition) and sort them
  //             by the LOW lb key bits; each (tile, bin) run reserves its slot in the bucket with
  //             one global atomic. The source block of an element (needed to rebuild its global
  //             scalar index) is found by a binary search in the partition's per-block offset row.
  // Output is what bucket accumulation consumes: count[], offs[], sorted[] (point index | sign<<31).
  struct SortPlan {
    int hb, lb;       // high / low bucket-key bits, hb + lb = c - 1
    int jb;           // bits for the precompute index j
    int chunk_log;    // scalars per pass-A block = 2^chunk_log
    int nblk;         // pass-A blocks
  };

  // digit word: |d| | (d<0)<<31, |d| in [0, 2^(c-1)], 0 = skip. Signed recoding as cpu_msm.hpp:289-295.
  // Windows are consumed in order, lowest first: the scalar is shifted right by c after each digit, so every
  // register index is static (a dynamically indexed w[] lives in scratch memory: 44 B/lane and the kernel waited
  // on it for 83 % of its cycles).
  struct DigitIter {
    uint32_t w[9];
    uint32_t carry = 0;
    __device__ __forceinline__ uint32_t next(int c)
    {
      uint32_t v = (w[0] & ((1u << c) - 1)) + carry;
#pragma unroll
      for (int k = 0; k < 8; k++)
        w[k] = __funnelshift_r(w[k], w[k + 1], c); // (w[k+1]:w[k]) >> c, c < 32
      const uint32_t half = 1u << (c - 1);
      if (v > half) {
        carry = 1;
        const uint32_t d = (1u << c) - v;
        return d ? (d | 0x80000000u) : 0u;
      }
      carry = 0;
      return v;
    }
  };

  // `bits`: scalar bits the MSM considers (MSMConfig.bitsize). The reference reads ONLY those bits of a scalar
  // (cpu_msm.hpp:288 coeff_width = min(c, bitsize - offset)), i.e. it computes sum (s_i mod 2^bitsize) P_i even when a
  // scalar is wider than the caller promised (icicle/tests/test_curve_api.cpp:82-123 msm_bitsize does exactly that).
  template <class C>
  __device__ __forceinline__ void load_scalar(DigitIter& it, const uint32_t* __restrict__ scalars, size_t i, bool scalars_refmont, int bits)
  {
    using FR = FieldOps<typename C::fr>;
    static_assert(FR::N32 == 8, "scalar fields here are 8 x u32");
    const uint4* p = reinterpret_cast<const uint4*>(scalars + i * 8);
    const uint4 lo = p[0], hi = p[1];
    it.w[0] = lo.x, it.w[1] = lo.y, it.w[2] = lo.z, it.w[3] = lo.w;
    it.w[4] = hi.x, it.w[5] = hi.y, it.w[6] = hi.z, it.w[7] = hi.w;
    if (scalars_refmont) { // x*2^256 -> x  (cpu_msm.hpp:274-275 from_montgomery)
      typename FR::fe cst;
#pragma unroll
      for (int k = 0; k < FR::N; k++)
        cst.l[k] = C::fr::REFMONT_TO_CANON[k];
      BF_SET_BOUND(cst, 1);
      FR::pack(it.w, FR::reduce(FR::mul(FR::unpack(it.w), cst)));
    }
    if (bits < 256) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int lo = 32 * k;
        if (bits <= lo)
          it.w[k] = 0;
        else if (bits < lo + 32)
          it.w[k] &= (1u << (bits - lo)) - 1u;
      }
    }
    it.w[8] = 0;
    it.carry = 0;
  }
  // window widths of a launch (msm_plan.h MsmPlan): windows [0, n_lo) are c - 1 bits wide, the others c; `negate`: a scalar
  // with its top bit set is replaced by r - s and every digit of it changes sign (cpu_msm.hpp:276-277,300-306)
  struct WinWidths {
    int c, n_lo;
    bool negate;
    __host__ __device__ int width(int wi) const { return wi < n_lo ? c - 1 : c; }
  };
  template <class C>
  __device__ __forceinline__ bool negate_scalar_if_top_bit(DigitIter& it)
  {
    constexpr int TOP = C::fr::NBITS - 1;
    if (!((it.w[TOP >> 5] >> (TOP & 31)) & 1u)) return false;
    uint32_t borrow = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { // r - s, both below 2^256
      const uint64_t d = (uint64_t)C::fr::P32[k] - it.w[k] - borrow;
      it.w[k] = (uint32_t)d;
      borrow = (uint32_t)(d >> 63);
    }
    return true;
  }

  // digits of all windows, dig[wi*n + i] (coalesced 4-byte writes; read back window by window)
  template <class C>
  __global__ __launch_bounds__(256) void k_digits(const uint32_t* __restrict__ scalars, uint32_t* __restrict__ dig, int n, size_t nscal, WinWidths ww, int nwin, bool scalars_refmont, int bits)
  {
    // nscal = (MSMs in this launch) * n scalars; row (b*nwin + wi) of `dig` holds window wi of MSM b
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nscal) return;
    const size_t b = t / n, i = t - b * n;
    DigitIter it;
    load_scalar<C>(it, scalars, t, scalars_refmont, bits);
    const uint32_t flip = (ww.negate && negate_scalar_if_top_bit<C>(it)) ? 0x80000000u : 0u;
    for (int wi = 0; wi < nwin; wi++) {
      const uint32_t d = it.next(ww.width(wi));
      dig[(b * nwin + wi) * n + i] = d ? (d ^ flip) : 0u;
    }
  }

  // k_digits fused with pass A's histogram (k_a_count): block (b, mb) owns scalar chunk b of MSM mb, writes its digits
  // and counts, per target window, how many fall into each of the 2^hb partitions -- the counts come from registers
  // instead of a second read of the 13 x 4 B per scalar digit array. Dynamic LDS: wpf * 2^hb counters.
  template <class C>
  __global__ __launch_bounds__(1024) void k_digits_count(const uint32_t* __restrict__ scalars, uint32_t* __restrict__ dig, uint32_t* __restrict__ cntA, int n, WinWidths ww, int nwin, int wpf, SortPlan sp, bool scalars_refmont, int bits)
  {
    extern __shared__ uint32_t lds[];
    const int b = blockIdx.x, mb = blockIdx.y;
    const uint32_t D = 1u << sp.hb, nh = (uint32_t)wpf * D;
    for (uint32_t k = threadIdx.x; k < nh; k += blockDim.x)
      lds[k] = 0;
    __syncthreads();
    const int lo = b << sp.chunk_log, hi = min(n, lo + (1 << sp.chunk_log));
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      DigitIter it;
      load_scalar<C>(it, scalars, (size_t)mb * n + i, scalars_refmont, bits);
      const uint32_t flip = (ww.negate && negate_scalar_if_top_bit<C>(it)) ? 0x80000000u : 0u;
      int wp = 0;
      for (int wi = 0; wi < nwin; wi++) {
        uint32_t d = it.next(ww.width(wi));
        d = d ? (d ^ flip) : 0u;
        dig[((size_t)mb * nwin + wi) * n + i] = d;
        const uint32_t key = d & 0x7fffffffu;
        if (key) atomicAdd(&lds[(uint32_t)wp * D + ((key - 1) >> sp.lb)], 1u);
        if (++wp == wpf) wp = 0;
      }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nh; k += blockDim.x) {
      const uint32_t wp = k >> sp.hb, h = k & (D - 1);
      cntA[((((size_t)mb * wpf + wp) << sp.hb) + h) * sp.nblk + b] = lds[k];
    }
  }

  // exclusive prefix of one value per thread over the block (blockDim.x a multiple of 64, <= 1024);
  // wsum: >= 17 words of LDS scratch
  __device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* wsum)
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    if (wave == 0) {
      const uint32_t t = lane < nwv ? wsum[lane] : 0;
      uint32_t sft = t;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t y = __shfl_up(sft, d);
        if (lane >= d) sft += y;
      }
      if (lane < nwv) wsum[lane] = sft - t;
    }
    __syncthreads();
    const uint32_t r = x - v + wsum[wave];
    __syncthreads();
    return r;
  }

  // Tile sort in LDS: a block ranks TS = 16 * blockDim elements by destination, stages them sorted,
  // and writes each destination's run contiguously -- HBM sees >= 64-byte runs instead of 4-byte
  // scatters (the unstaged scatter wrote 30 GB to place 3.5 GB, profiles/r01_notes.md).
  constexpr int SORT_EPT = 16;                 // elements per thread per tile
  constexpr uint32_t SORT_TS = 1024 * SORT_EPT; // tile size with 1024 threads
  // Two block shapes. 1024 threads (a whole CU: 16 waves, ~100 KiB of LDS) is the stand-alone sort. 256 threads (one wave per
  // SIMD, a 4096-element tile, ~36 KiB) is the CO-RESIDENT sort of the pipelined schedule (msm_run_single, window groups):
  // its waves fit beside two k_accumulate waves per SIMD, so the sort of window group g + 1 -- waves parked on memory most
  // of the time -- runs in the issue slots the accumulation of group g leaves, instead of before it.
  struct TileLds {
    uint32_t* cnt;   // [D] per-destination count of this tile -> reused as tile-local offset
    uint32_t* gbase; // [D] global position of this tile's run per destination
    uint32_t* stage; // [TS]
    uint16_t* sdest; // [TS]
    uint32_t* wsum;  // [32]
  };
  __device__ __forceinline__ TileLds tile_lds(uint32_t* lds, uint32_t D, uint32_t TS = SORT_TS)
  {
    TileLds t;
    t.cnt = lds;
    t.gbase = lds + D;
    t.stage = lds + 2 * D;
    t.sdest = reinterpret_cast<uint16_t*>(lds + 2 * D + TS);
    t.wsum = lds + 2 * D + TS + TS / 2;
    return t;
  }
  __host__ __device__ static inline size_t tile_lds_bytes(uint32_t D, uint32_t TS = SORT_TS) { return ((size_t)2 * D + TS + TS / 2 + 32) * 4; }

  // pass A count: block (b, wl) histograms the high key bits of scalar chunk b for target window w0+wl
  static __global__ __launch_bounds__(1024) void k_a_count(const uint32_t* __restrict__ dig, uint32_t* __restrict__ cntA, int n, int nwin, int wpf, int pf, SortPlan sp)
  {
    // wl = (MSM index in this launch) * wpf + target window
    extern __shared__ uint32_t lds[];
    const int b = blockIdx.x, wl = blockIdx.y, wp = wl % wpf;
    const size_t rowbase = (size_t)(wl / wpf) * nwin;
    const uint32_t D = 1u << sp.hb;
    for (uint32_t k = threadIdx.x; k < D; k += blockDim.x)
      lds[k] = 0;
    __syncthreads();
    const int lo = b << sp.chunk_log, hi = min(n, lo + (1 << sp.chunk_log));
    for (int j = 0; j < pf; j++) {
      const int wi = j * wpf + wp;
      if (wi >= nwin) break;
      const uint32_t* d = dig + (rowbase + wi) * n;
      for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 8 * blockDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { // all loads first: the histogram update depends on each of them
          const int i = i0 + u * (int)blockDim.x;
          v[u] = i < hi ? d[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const uint32_t key = v[u] & 0x7fffffffu;
          if (key) atomicAdd(&lds[(key - 1) >> sp.lb], 1u);
        }
      }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < D; k += blockDim.x)
      cntA[(((size_t)wl << sp.hb) + k) * sp.nblk + b] = lds[k];
  }

  // pass A scatter: element = sign | low key bits | j | index within chunk, into partition runs
  // FINAL: single-level sort (lb == 0): the element is already the bucket-list entry (point index | sign)
  template <bool FINAL, int TPB>
  __global__ __launch_bounds__(TPB) void k_a_scatter(const uint32_t* __restrict__ dig, const uint32_t* __restrict__ offA, uint32_t* __restrict__ outA, int n, int nwin, int wpf, int pf, SortPlan sp, size_t cap, int wl0)
  {
    // wl0: first window of this launch (window groups of the pipelined schedule); every table is indexed by the global wl
    extern __shared__ uint32_t lds[];
    constexpr uint32_t TS = (uint32_t)TPB * SORT_EPT;
    const int b = blockIdx.x, wl = wl0 + (int)blockIdx.y, wp = wl % wpf;
    const size_t rowbase = (size_t)(wl / wpf) * nwin;
    const uint32_t D = 1u << sp.hb;
    const uint32_t R = (D + TPB - 1) / TPB, k0 = threadIdx.x * R; // destinations [k0, k0 + R) belong to this thread (D <= TPB: one)
    TileLds t = tile_lds(lds, D, TS);
    uint32_t* cursor = lds + tile_lds_bytes(D, TS) / 4; // [D] running write position per destination
    for (uint32_t k = threadIdx.x; k < D; k += TPB)
      cursor[k] = offA[(((size_t)wl << sp.hb) + k) * sp.nblk + b];
    __syncthreads();
    const int lo = b << sp.chunk_log, hi = min(n, lo + (1 << sp.chunk_log));
    const uint32_t lmask = (1u << sp.lb) - 1;
    uint32_t* dst = outA + (size_t)wl * cap;
    for (int j = 0; j < pf; j++) {
      const int wi = j * wpf + wp;
      if (wi >= nwin) break;
      const uint32_t* d = dig + (rowbase + wi) * n;
      // the digit words of the NEXT tile are fetched before this tile is ranked / staged / written, so the
      // HBM reads overlap the LDS phases (the kernel used to be parked on s_waitcnt for 81 % of its cycles)
      auto fetch = [&](int tile0, uint32_t* buf) {
#pragma unroll
        for (int it = 0; it < SORT_EPT; it++) {
          const int i = tile0 + it * TPB + threadIdx.x;
          buf[it] = i < hi ? d[i] : 0u;
        }
      };
      uint32_t cur[SORT_EPT];
      fetch(lo, cur);
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the first tile lands before the loop (see k_ntt_fast: a pending entry edge makes the waitcnt pass guard every iteration conservatively)
      for (int tile0 = lo; tile0 < hi; tile0 += TS) {
        uint32_t nxt[SORT_EPT];
        fetch(tile0 + (int)TS, nxt); // past the end of the chunk every lane is predicated off (zeros)
        for (uint32_t k = threadIdx.x; k < D; k += TPB)
          t.cnt[k] = 0;
        __syncthreads();
        uint32_t el[SORT_EPT], dr[SORT_EPT]; // element, (dest << 16 | rank)
#pragma unroll
        for (int it = 0; it < SORT_EPT; it++) {
          const int i = tile0 + it * TPB + threadIdx.x;
          dr[it] = 0xffffffffu;
          if (i < hi) {
            const uint32_t dv = cur[it];
            const uint32_t key = dv & 0x7fffffffu;
            if (key) {
              const uint32_t km = key - 1, h = km >> sp.lb;
              el[it] = FINAL ? (((uint32_t)i * (uint32_t)pf + (uint32_t)j) | (dv & 0x80000000u))
                             : ((dv & 0x80000000u) | ((km & lmask) << (31 - sp.lb)) | ((uint32_t)j << (31 - sp.lb - sp.jb)) | (uint32_t)(i - lo));
              dr[it] = (h << 16) | atomicAdd(&t.cnt[h], 1u);
            }
          }
        }
        __syncthreads();
        uint32_t mysum = 0;
        for (uint32_t r = 0; r < R; r++)
          if (k0 + r < D) mysum += t.cnt[k0 + r];
        uint32_t toff = block_exscan(mysum, t.wsum);
        for (uint32_t r = 0; r < R; r++)
          if (k0 + r < D) {
            const uint32_t c1 = t.cnt[k0 + r];
            t.cnt[k0 + r] = toff; // now the tile-local offset
            t.gbase[k0 + r] = cursor[k0 + r];
            cursor[k0 + r] += c1;
            toff += c1;
          }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < SORT_EPT; it++)
          if (dr[it] != 0xffffffffu) {
            const uint32_t h = dr[it] >> 16, pos = t.cnt[h] + (dr[it] & 0xffffu);
            t.stage[pos] = el[it];
            t.sdest[pos] = (uint16_t)h;
          }
        __syncthreads();
        // the next tile's words move into `cur` BEFORE this tile's stores are issued: a wait for them at the top of
        // the next iteration would be a vmcnt(0) behind a data-dependent number of stores, i.e. it would also wait
        // for every store of this tile (one exposed store latency per tile with a single 1024-thread block per CU)
        // (the empty asm pins the point where the loaded value must exist: a plain copy is a PHI and its moves -- with
        // their vmcnt(0) -- are materialised in the loop latch, after the stores)
#pragma unroll
        for (int it = 0; it < SORT_EPT; it++) {
          cur[it] = nxt[it];
          asm volatile("" : "+v"(cur[it]));
        }
        const uint32_t ntile = t.cnt[D - 1] + (cursor[D - 1] - t.gbase[D - 1]);
        for (uint32_t sidx = threadIdx.x; sidx < ntile; sidx += TPB) {
          const uint32_t h = t.sdest[sidx];
          dst[t.gbase[h] + (sidx - t.cnt[h])] = t.stage[sidx];
        }
        __syncthreads();
      }
    }
  }

  // Row-wise exclusive scans (pass-A counter tables [2^hb][nblk] per window, bucket counts per window).
  // Two launches over (chunk, row): chunk sums, then each chunk scans itself on top of the sums before it --
  // a single block per row left 243 of 256 CUs idle and took ~1 ms per table at 2^26.
  constexpr uint32_t SCAN_CHUNK = 8192; // 1024 threads x 8 consecutive elements
  static __global__ __launch_bounds__(1024) void k_scan_sums(const uint32_t* __restrict__ in, uint32_t* __restrict__ sums, uint32_t m)
  {
    __shared__ uint32_t wsum[32];
    const uint32_t c = blockIdx.x, row = blockIdx.y, nchunks = gridDim.x;
    const uint32_t* src = in + (size_t)row * m;
    const uint32_t k0 = c * SCAN_CHUNK + threadIdx.x * 8;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
      v += (k0 + j < m) ? src[k0 + j] : 0u;
    const uint32_t ex = block_exscan(v, wsum);
    if (threadIdx.x == 1023) sums[(size_t)row * nchunks + c] = ex + v;
  }
  static __global__ __launch_bounds__(1024) void k_scan_apply(const uint32_t* __restrict__ in, const uint32_t* __restrict__ sums, uint32_t* __restrict__ out, uint32_t* __restrict__ out2, uint32_t* __restrict__ totals, uint32_t m)
  {
    __shared__ uint32_t wsum[32];
    __shared__ uint32_t sbase;
    const uint32_t c = blockIdx.x, row = blockIdx.y, nchunks = gridDim.x;
    // sum of the chunks before this one
    uint32_t part = 0;
    for (uint32_t q = threadIdx.x; q < c; q += blockDim.x)
      part += sums[(size_t)row * nchunks + q];
    const uint32_t pex = block_exscan(part, wsum);
    if (threadIdx.x == 1023) sbase = pex + part;
    __syncthreads();
    const uint32_t base = sbase;
    const uint32_t* src = in + (size_t)row * m;
    const uint32_t k0 = c * SCAN_CHUNK + threadIdx.x * 8;
    uint32_t x[8], v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x[j] = (k0 + j < m) ? src[k0 + j] : 0u;
      v += x[j];
    }
    uint32_t run = base + block_exscan(v, wsum);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (k0 + j < m) {
        out[(size_t)row * m + k0 + j] = run;
        if (out2) out2[(size_t)row * m + k0 + j] = run;
      }
      run += x[j];
    }
    if (totals && c == nchunks - 1 && threadIdx.x == 1023) totals[row] = run; // row total
  }

  // single-level sort (lb == 0): bucket k of window wl IS partition k; count/offs come from pass A's table
  static __global__ __launch_bounds__(256) void k_tables_from_a(const uint32_t* __restrict__ offA, uint32_t* __restrict__ count, uint32_t* __restrict__ offs, size_t nbk, uint32_t nb, int nblk, size_t totals_base, size_t bk0)
  {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nbk) return;
    t += bk0;
    const size_t wl = t / nb;
    const uint32_t k = (uint32_t)(t - wl * nb);
    const uint32_t ps = offA[t * nblk];
    const uint32_t pe = (k + 1 < nb) ? offA[(t + 1) * nblk] : offA[totals_base + wl];
    offs[t] = ps;
    count[t] = pe - ps;
  }

  // pass B. A partition (window wp, high key bits h) is cut into sub-chunks of CHUNKB elements, one
  // block each, so neither a sparse top window nor skewed scalars can serialise it on one block.
  // k_b_plan turns partition sizes into a block -> (partition, sub-chunk) table; k_b_count adds LDS
  // histograms of the low lb key bits into count[]; k_scan_buckets makes offs[]/cursor[];
  // k_b_scatter reserves a range per (block, bin) with ONE global atomic and ranks inside it in LDS.
  constexpr uint32_t CHUNKB_LOG = 17;

  static __global__ __launch_bounds__(1024) void k_b_plan(const uint32_t* __restrict__ offA, uint32_t* __restrict__ bstart, uint32_t nparts, int wpf, int hb, int nblk, uint32_t p0)
  {
    // plans partitions [p0, p0 + nparts) (one window group); bstart[] is local to the group, wpf = windows of the whole launch
    __shared__ uint32_t part[1024];
    const uint32_t per = (nparts + 1023) / 1024;
    const uint32_t lo = min(nparts, threadIdx.x * per), hi = min(nparts, lo + per);
    const uint32_t nparts_w = 1u << hb;
    auto psize = [&](uint32_t pl) -> uint32_t {
      const uint32_t p = p0 + pl;
      const uint32_t wp = p >> hb, h = p & (nparts_w - 1);
      const size_t row = (size_t)p * nblk;
      const uint32_t ps = offA[row];
      const uint32_t pe = (h + 1 < nparts_w) ? offA[row + nblk] : offA[(size_t)wpf * nparts_w * nblk + wp];
      return pe - ps;
    };
    uint32_t s = 0;
    for (uint32_t p = lo; p < hi; p++)
      s += (psize(p) + (1u << CHUNKB_LOG) - 1) >> CHUNKB_LOG;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t v = (threadIdx.x >= (unsigned)d) ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t p = lo; p < hi; p++) {
      bstart[p] = run;
      run += (psize(p) + (1u << CHUNKB_LOG) - 1) >> CHUNKB_LOG;
    }
    if (threadIdx.x == 1023) bstart[nparts] = part[1023];
  }

  // block -> (partition p, element range [r0,r1) in the window's pass-A array); false if idle
  __device__ __forceinline__ bool b_locate(const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ offA, uint32_t nparts, int wpf, int hb, int nblk, uint32_t p0, uint32_t& p, uint32_t& r0, uint32_t& r1)
  {
    const uint32_t blk = blockIdx.x;
    if (blk >= bstart[nparts]) return false;
    uint32_t lo = 0, hi = nparts; // last (group-local) p with bstart[p] <= blk
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (bstart[mid] <= blk) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    const uint32_t first = bstart[lo];
    p = p0 + lo; // global partition index from here on
    const uint32_t nparts_w = 1u << hb;
    const uint32_t wp = p >> hb, h = p & (nparts_w - 1);
    const size_t row = (size_t)p * nblk;
    const uint32_t ps = offA[row];
    const uint32_t pe = (h + 1 < nparts_w) ? offA[row + nblk] : offA[(size_t)wpf * nparts_w * nblk + wp];
    r0 = ps + ((blk - first) << CHUNKB_LOG);
    r1 = min(pe, r0 + (1u << CHUNKB_LOG));
    return r0 < r1;
  }

  static __global__ __launch_bounds__(1024) void k_b_count(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ offA, const uint32_t* __restrict__ bstart, uint32_t* __restrict__ count, uint32_t nparts, int wpf, SortPlan sp, size_t cap, uint32_t nb, uint32_t p0)
  {
    extern __shared__ uint32_t lds[];
    uint32_t p, r0, r1;
    if (!b_locate(bstart, offA, nparts, wpf, sp.hb, sp.nblk, p0, p, r0, r1)) return;
    const uint32_t nbins = 1u << sp.lb;
    for (uint32_t k = threadIdx.x; k < nbins; k += blockDim.x)
      lds[k] = 0;
    __syncthreads();
    const uint32_t wp = p >> sp.hb, h = p & ((1u << sp.hb) - 1);
    const uint32_t* src = inA + (size_t)wp * cap;
    const int lshift = 31 - sp.lb;
    const uint32_t lmask = nbins - 1;
    for (uint32_t p0 = r0 + threadIdx.x; p0 < r1; p0 += 8 * blockDim.x) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t pos = p0 + u * blockDim.x;
        v[u] = pos < r1 ? src[pos] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (p0 + u * blockDim.x < r1) atomicAdd(&lds[(v[u] >> lshift) & lmask], 1u);
    }
    __syncthreads();
    uint32_t* cw = count + (size_t)wp * nb + ((size_t)h << sp.lb);
    for (uint32_t k = threadIdx.x; k < nbins; k += blockDim.x)
      if (lds[k]) atomicAdd(&cw[k], lds[k]);
  }

  template <int TPB>
  __global__ __launch_bounds__(TPB) void k_b_scatter(const uint32_t* __restrict__ inA, const uint32_t* __restrict__ offA, const uint32_t* __restrict__ bstart, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted, uint32_t nparts, int wpf, int pf, SortPlan sp, size_t cap, uint32_t nb, uint32_t p0)
  {
    extern __shared__ uint32_t lds[]; // tile-sort arrays | [nblk+1] piece offsets of this partition
    constexpr uint32_t TS = (uint32_t)TPB * SORT_EPT;
    uint32_t p, r0, r1;
    if (!b_locate(bstart, offA, nparts, wpf, sp.hb, sp.nblk, p0, p, r0, r1)) return;
    const uint32_t D = 1u << sp.lb;
    const uint32_t R = (D + TPB - 1) / TPB, k0 = threadIdx.x * R; // bins [k0, k0 + R) belong to this thread
    TileLds t = tile_lds(lds, D, TS);
    uint32_t* boffs = lds + tile_lds_bytes(D, TS) / 4;
    uint32_t* gmap = boffs + sp.nblk + 1; // [TS / 64]
    const uint32_t wp = p >> sp.hb, h = p & ((1u << sp.hb) - 1);
    const uint32_t nparts_w = 1u << sp.hb;
    const size_t row = (size_t)p * sp.nblk;
    for (uint32_t k = threadIdx.x; k <= (uint32_t)sp.nblk; k += TPB)
      boffs[k] = (k < (uint32_t)sp.nblk) ? offA[row + k] : ((h + 1 < nparts_w) ? offA[row + sp.nblk] : offA[(size_t)wpf * nparts_w * sp.nblk + wp]);
    __syncthreads();
    const uint32_t* src = inA + (size_t)wp * cap;
    uint32_t* dst = sorted + (size_t)wp * cap;
    uint32_t* cw = cursor + (size_t)wp * nb + ((size_t)h << sp.lb);
    const int lshift = 31 - sp.lb;
    const uint32_t lmask = D - 1;
    const uint32_t imask = (1u << (31 - sp.lb - sp.jb)) - 1;
    const uint32_t jmask = (1u << sp.jb) - 1;
    auto fetch = [&](uint32_t tile0, uint32_t* buf) { // next tile's elements, see k_a_scatter
#pragma unroll
      for (int it = 0; it < SORT_EPT; it++) {
        const uint32_t pos = tile0 + it * TPB + threadIdx.x;
        buf[it] = pos < r1 ? src[pos] : 0u;
      }
    };
    uint32_t cur[SORT_EPT];
    fetch(r0, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), see k_a_scatter
    for (uint32_t tile0 = r0; tile0 < r1; tile0 += TS) {
      uint32_t nxt[SORT_EPT];
      fetch(tile0 + TS, nxt); // past the end of the run every lane is predicated off (zeros)
      for (uint32_t k = threadIdx.x; k < D; k += TPB) // (D = 2048 bins at c = 22)
        t.cnt[k] = 0;
      // source block (scalar chunk) of an element = last bsrc with boffs[bsrc] <= pos. One binary search per
      // 64 consecutive positions (once per tile); an element then starts from its group's answer
      // and walks forward (pieces are ~64 elements long, so 0-2 steps) instead of 10 dependent LDS reads each.
      if (threadIdx.x < TS / 64) {
        const uint32_t pos = tile0 + threadIdx.x * 64;
        uint32_t blo = 0, bhi = sp.nblk;
        if (pos < r1) {
          while (bhi - blo > 1) {
            const uint32_t mid = (blo + bhi) >> 1;
            if (boffs[mid] <= pos) {
              blo = mid;
            } else {
              bhi = mid;
            }
          }
        }
        gmap[threadIdx.x] = blo;
      }
      __syncthreads();
      uint32_t el[SORT_EPT], dr[SORT_EPT];
#pragma unroll
      for (int it = 0; it < SORT_EPT; it++) {
        const uint32_t pos = tile0 + it * TPB + threadIdx.x;
        dr[it] = 0xffffffffu;
        if (pos < r1) {
          const uint32_t e = cur[it];
          uint32_t blo = gmap[(it * TPB + threadIdx.x) >> 6];
          while (blo + 1 < (uint32_t)sp.nblk && boffs[blo + 1] <= pos)
            blo++;
          const uint32_t i = (blo << sp.chunk_log) + (e & imask);
          const uint32_t j = (e >> (31 - sp.lb - sp.jb)) & jmask;
          const uint32_t bin = (e >> lshift) & lmask;
          el[it] = (i * (uint32_t)pf + j) | (e & 0x80000000u);
          dr[it] = (bin << 16) | atomicAdd(&t.cnt[bin], 1u);
        }
      }
      __syncthreads();
      {
        uint32_t mysum = 0;
        for (uint32_t r = 0; r < R; r++)
          if (k0 + r < D) mysum += t.cnt[k0 + r];
        uint32_t toff = block_exscan(mysum, t.wsum);
        for (uint32_t r = 0; r < R; r++)
          if (k0 + r < D) {
            const uint32_t c1 = t.cnt[k0 + r];
            t.cnt[k0 + r] = toff;
            t.gbase[k0 + r] = c1 ? atomicAdd(&cw[k0 + r], c1) : 0u; // reserve the run in the bucket list
            toff += c1;
          }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < SORT_EPT; it++)
        if (dr[it] != 0xffffffffu) {
          const uint32_t bin = dr[it] >> 16, pos = t.cnt[bin] + (dr[it] & 0xffffu);
          t.stage[pos] = el[it];
          t.sdest[pos] = (uint16_t)bin;
        }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < SORT_EPT; it++) { // before the stores, pinned: see k_a_scatter
        cur[it] = nxt[it];
        asm volatile("" : "+v"(cur[it]));
      }
      const uint32_t ntile = min(r1 - tile0, TS);
      for (uint32_t sidx = threadIdx.x; sidx < ntile; sidx += TPB) {
        const uint32_t bin = t.sdest[sidx];
        dst[t.gbase[bin] + (sidx - t.cnt[bin])] = t.stage[sidx];
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------------------------------
  // 4. bucket accumulation. Thread t < nbk owns bucket t and accumulates its first `seg` points
  //    (XYZZ accumulator in registers); a bucket with more points (sparse top window, skewed scalars:
  //    wrappers/rust/icicle-core/src/msm/tests.rs:256-304) gets overflow segments of `seg` points each,
  //    planned by k_plan_overflow, accumulated by threads t >= nbk and folded in by k_fold_overflow.
  struct OvfSeg {
    uint32_t bucket; // global bucket id
    uint32_t start;  // first point of this segment inside the bucket
    uint32_t first;  // 1 if this is the first overflow segment of its bucket
    uint32_t nextra; // number of overflow segments of the bucket (valid when first)
  };

  // ovf_count[0] = overflow segments, [1] = overflowing buckets, [2] = largest number of segments of one bucket;
  // firsts[i] = slot of the first segment of the i-th overflowing bucket
  static __global__ __launch_bounds__(256) void k_plan_overflow(const uint32_t* __restrict__ count, size_t nbk, uint32_t seg, uint32_t* __restrict__ ovf_count, OvfSeg* __restrict__ ovf, uint32_t* __restrict__ firsts, uint32_t ovf_cap, size_t bk0)
  {
    // buckets [bk0, bk0 + nbk) (one window group); bucket ids in the plan are global
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nbk) return;
    t += bk0;
    const uint32_t cnt = count[t];
    if (cnt <= seg) return;
    const uint32_t extra = (cnt + seg - 1) / seg - 1;
    const uint32_t slot = atomicAdd(ovf_count, extra);
    if (slot < ovf_cap) {
      firsts[atomicAdd(ovf_count + 1, 1u)] = slot;
      atomicMax(ovf_count + 2, extra);
    }
    for (uint32_t sgi = 0; sgi < extra && slot + sgi < ovf_cap; sgi++) {
      OvfSeg o;
      o.bucket = (uint32_t)t;
      o.start = (sgi + 1) * seg;
      o.first = (sgi == 0);
      o.nextra = extra;
      ovf[slot + sgi] = o;
    }
  }

  // Size-balanced scheduling. A wave of k_accumulate runs until its longest bucket is done: with ~128 points
  // per bucket (Poisson, sigma 11) the longest of 64 is ~155, i.e. ~15 % of the lane-cycles of the dominant
  // kernel were idle (SQ_INSTS_VALU per mixed add 2912 vs ~2540 in the instruction stream). perm[] lists the
  // buckets by descending size class, so the 64 lanes of a wave get buckets of (almost) the same length and the
  // empty buckets end up in waves that exit at once. Counting sort over 257 classes of min(count, seg).
  constexpr uint32_t SZ_BINS = 257;
  constexpr uint32_t SZ_CHUNK = 16384; // buckets per block (1024 threads x 16)
  __device__ __forceinline__ uint32_t size_class(uint32_t cnt, uint32_t seg)
  {
    const uint32_t q = (uint32_t)(((uint64_t)min(cnt, seg) * 256u) / seg); // 0..256
    return 256u - q;                                                       // heavy first
  }
  // (mixed-width plans: the narrow windows [0, n_lo) use the first nb / 2 buckets of their slot; the other half is not a
  //  bucket -- it gets no thread of k_accumulate and nothing reads it. `used_bucket` on a global bucket id.)
  __device__ __forceinline__ bool used_bucket(size_t bucket, uint32_t nb, uint32_t n_lo)
  {
    return n_lo == 0 || bucket / nb >= n_lo || (bucket & (nb - 1)) < nb / 2;
  }
  static __global__ __launch_bounds__(1024) void k_bsize_count(const uint32_t* __restrict__ count, uint32_t* __restrict__ table, size_t nbk, uint32_t seg, size_t bk0, uint32_t nb, uint32_t n_lo)
  {
    __shared__ uint32_t hist[SZ_BINS];
    for (uint32_t k = threadIdx.x; k < SZ_BINS; k += blockDim.x)
      hist[k] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SZ_CHUNK;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const size_t b = base + it * 1024 + threadIdx.x;
      if (b < nbk && used_bucket(bk0 + b, nb, n_lo)) atomicAdd(&hist[size_class(count[bk0 + b], seg)], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < SZ_BINS; k += blockDim.x)
      table[(size_t)k * gridDim.x + blockIdx.x] = hist[k];
  }
  static __global__ __launch_bounds__(1024) void k_bsize_scatter(const uint32_t* __restrict__ count, const uint32_t* __restrict__ table_off, uint32_t* __restrict__ perm, size_t nbk, uint32_t seg, size_t bk0, uint32_t nb, uint32_t n_lo)
  {
    __shared__ uint32_t cursor[SZ_BINS];
    for (uint32_t k = threadIdx.x; k < SZ_BINS; k += blockDim.x)
      cursor[k] = table_off[(size_t)k * gridDim.x + blockIdx.x];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SZ_CHUNK;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const size_t b = base + it * 1024 + threadIdx.x;
      if (b < nbk && used_bucket(bk0 + b, nb, n_lo)) perm[atomicAdd(&cursor[size_class(count[bk0 + b], seg)], 1u)] = (uint32_t)(bk0 + b);
    }
  }

  // PAD: the kernel claims 176 VGPRs instead of the 155 it needs, so that exactly TWO of its waves fit a SIMD (3 x 176 > 512)
  // and 160 registers per lane stay free for a co-resident wave of the next window group's sort (pipelined schedule, off by
  // default). Really at two waves per SIMD the kernel is ~4.5 % SLOWER than at three (profiles/r05_notes.md section 3b; round 3's
  // "60.14 vs 60.10 ms" compared launch bounds of a kernel whose 155 VGPRs let three waves in either way): the default launch is MINW = 3.
  template <class C, int MINW, bool PAD = false>
  __global__ __launch_bounds__(128, MINW) void k_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ count, const uint32_t* __restrict__ offs, typename EC<C>::Proj* __restrict__ buckets, typename EC<C>::Proj* __restrict__ ovf_part, const OvfSeg* __restrict__ ovf, const uint32_t* __restrict__ ovf_count, const uint32_t* __restrict__ perm, uint32_t ovf_cap, uint32_t nb, size_t nbk, size_t cap, uint32_t seg, int wpf, size_t bases_stride)
  {
    // bases_stride: words between the base arrays of consecutive MSMs of a batch (0 = shared bases)
    // perm: thread t < nbk accumulates bucket perm[t] (size-balanced order)
    using E = EC<C>;
    constexpr int PW = 2 * E::N32; // words per affine point
    if constexpr (PAD) asm volatile("" ::: "v175");
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t bucket;
    uint32_t start;
    typename E::Proj* dst;
    // overflow segments first: each is a full `seg` points, the heaviest work items of the launch, so they must
    // not be left for the tail (unused overflow slots exit at once)
    if (t >= ovf_cap) {
      if (t - ovf_cap >= nbk) return;
      bucket = perm[t - ovf_cap];
      start = 0;
      dst = buckets + bucket;
    } else {
      const size_t o = t;
      if (o >= *ovf_count) return;
      bucket = ovf[o].bucket;
      start = ovf[o].start;
      dst = ovf_part + o;
    }
    const size_t wp = bucket / nb; // window index within the launch = (MSM index) * wpf + target window
    bases += (wp / wpf) * bases_stride;
    const uint32_t total = count[bucket];
    const uint32_t cnt = min(total - min(total, start), seg);
    const uint32_t* src = sorted + wp * cap + offs[bucket] + start;
    typename E::XYZZ acc;
    bool empty = true;
    // (Prefetching the next point was measured twice in round 2 and bought nothing. In registers it costs 16 VGPRs that
    // 3 waves per SIMD do not have: spills, and every scratch reload's vmcnt(0) then waits for the prefetch. Through
    // LDS (global_load_lds_dwordx4 into a per-wave double buffer, no VGPRs, waits verified in the ISA) it is a real
    // one-iteration-ahead prefetch and still changes nothing: 60.0 vs 60.5 ms. The other two waves of the SIMD already
    // cover the gather latency; the kernel sits at the VALU issue roof -- profiles/r02_notes.md section 1.)
    // The list is read FOUR entries at a time, one group ahead: a lane's list is contiguous but no other lane shares
    // its 64-byte lines, and by the time a lane came back for the next 4-byte entry (one mixed add = ~4.5 us later)
    // the line had usually left the L2 -- each entry then cost a 64-byte fetch (PMC: 77 GB per 2^26 MSM against
    // 59 GB of gathers + entries, profiles/r02_notes.md). The reads may run up to 7 entries past the end of the
    // bucket's list (never used): `sorted` is allocated with that slack.
    auto load4 = [&](uint32_t j) {
      uint4 v;
      __builtin_memcpy(&v, src + j, 16); // 4-byte aligned
      return v;
    };
    uint4 cur = load4(0), nxt = cur;
    for (uint32_t j = 0; j < cnt; j++) {
      if ((j & 3u) == 0) nxt = load4(j + 4);
      const uint32_t e = cur.x;
      cur.x = cur.y, cur.y = cur.z, cur.z = cur.w;
      if ((j & 3u) == 3u) cur = nxt;
      const uint4* p = reinterpret_cast<const uint4*>(bases + (size_t)(e & 0x7fffffffu) * PW);
      uint32_t w[PW];
#pragma unroll
      for (int q = 0; q < PW / 4; q++) {
#ifdef MSM_NT_GATHER // A/B: keep the once-used base points out of the L2 (tools/ab_lib.sh)
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p) + q);
#else
        const uint4 v = p[q];
#endif
        w[4 * q] = v.x;
        w[4 * q + 1] = v.y;
        w[4 * q + 2] = v.z;
        w[4 * q + 3] = v.w;
      }
      if (__builtin_expect(E::words_are_zero(w), 0)) continue; // identity base: contributes nothing (cpu_msm.hpp:282)
      typename E::Aff a = E::cneg(E::load_plain(w), (e >> 31) != 0);
      E::madd(acc, empty, a);
    }
    *dst = E::to_proj(acc, empty);
  }

  // buckets[b] += its overflow partials. A group of G lanes handles one overflowing bucket: lanes fold strided
  // partials, then a tree through LDS, so a bucket with thousands of segments -- a 1-bit top window, all-equal
  // scalars -- costs log-depth, not a serial chain. G follows the largest segment count of the launch: every tree
  // level is a wave-wide complete addition whether 2 or 64 lanes hold something, so small groups win as long as the
  // serial part stays short. Measured at BN254 2^26 (short top window, 15-17 segments per bucket, 16 K buckets):
  // G = 64: 0.70 ms, G = 16: 0.21 ms, G = 4: 0.12 ms (profiles/r02_notes.md section 11).
  template <class C>
  __global__ __launch_bounds__(64) void k_fold_overflow(typename EC<C>::Proj* __restrict__ buckets, const typename EC<C>::Proj* __restrict__ ovf_part, const OvfSeg* __restrict__ ovf, const uint32_t* __restrict__ firsts, const uint32_t* __restrict__ ovf_count, uint32_t ovf_cap)
  {
    using E = EC<C>;
    __shared__ typename E::Proj sh[64];
    const uint32_t n = min(ovf_count[0], ovf_cap), nfirst = ovf_count[1];
    const uint32_t mx = ovf_count[2]; // largest number of overflow segments of one bucket
    const uint32_t G = mx <= 2 ? 1u : (mx <= 32 ? 4u : (mx <= 256 ? 16u : 64u)), per_block = 64 / G;
    const uint32_t lane = threadIdx.x, sub = lane / G, gl = lane % G;
    for (uint32_t q0 = blockIdx.x * per_block; q0 < nfirst; q0 += gridDim.x * per_block) {
      const uint32_t q = q0 + sub;
      const bool act = q < nfirst;
      const uint32_t o = act ? firsts[q] : 0;
      const uint32_t ne = act ? min(ovf[o].nextra, n - o) : 0;
      typename E::Proj v = E::proj_identity();
      for (uint32_t k = gl; k < ne; k += G)
        v = E::add(v, ovf_part[o + k]);
      sh[lane] = v;
      __syncthreads();
      for (uint32_t s = G / 2; s >= 1; s >>= 1) {
        if (gl < s) {
          v = E::add(v, sh[lane + s]);
          sh[lane] = v;
        }
        __syncthreads();
      }
      if (act && gl == 0) buckets[ovf[o].bucket] = E::add(buckets[ovf[o].bucket], v);
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------------------------------
  // 5a/5b. bucket reduction of a window, S = sum_k (k+1) * B_k, with wave64 scans.
  //
  // A chunk = 64*m consecutive buckets [k0, k0 + 64m) is owned by ONE WAVE, lane l taking buckets k0 + 64 i + l,
  // i < m -- so every step of the wave reads 64 consecutive buckets (coalesced), unlike a thread-per-segment walk.
  // With (k+1) = (k0+1) + 64 i + l and, per lane, line_l = sum_i B (column sum) and tri0_l = sum_i i*B (zero-based
  // running sum, two complete adds per bucket):
  //     S_chunk = (k0+1) * T + 64 * sum_l tri0_l + sum_l l * line_l ,   T = sum_l line_l
  // and sum_l l*line_l = sum_l X_l with X_l = sum_{l' > l} line_l' -- an exclusive SUFFIX SCAN across the lanes of the
  // wave (6 shuffle + add steps). So a wave emits V = sum_l (64 * tri0_l + X_l) (one more 6-step wave reduction) and T;
  // the per-window kernel repeats the same trick over chunks: S = sum_c (V_c + T_c) + 64m * sum_c c * T_c, the last
  // sum again a suffix scan, across the block. No scalar multiplication anywhere (round 1 paid a ~19-doubling
  // mul_small per 32-bucket segment), 2 + 13/m adds per bucket.
  // Only chunks [seg_lo, seg_lo + nsegr) of every window are reduced (the whole window unless a multi-device bucket
  // exchange left this device a slice); the chunk outputs are compact: [window][nsegr].
  template <class C>
  struct ReduceWindowLanes { // block size of the per-window kernel (its LDS holds one projective point per thread)
    static constexpr int value = (sizeof(typename EC<C>::Proj) * 256 <= 60 * 1024) ? 256 : 128;
  };
  template <class P>
  __device__ __forceinline__ P proj_shfl_down(const P& v, int d)
  {
    static_assert(sizeof(P) % 4 == 0, "projective point is a whole number of words");
    P r;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (size_t i = 0; i < sizeof(P) / 4; i++)
      dst[i] = __shfl_down(src[i], d);
    return r;
  }
  // inclusive suffix sum over the 64 lanes of a wave: lane l gets sum_{l' >= l} v_l'
  template <class C>
  __device__ __forceinline__ typename EC<C>::Proj wave_suffix_sum(typename EC<C>::Proj v, int lane)
  {
    using E = EC<C>;
    for (int d = 1; d < 64; d <<= 1) {
      const typename E::Proj o = proj_shfl_down(v, d);
      if (lane + d < 64) v = E::add(v, o);
    }
    return v;
  }
  // sum over the wave, valid in lane 0
  template <class C>
  __device__ __forceinline__ typename EC<C>::Proj wave_sum(typename EC<C>::Proj v, int lane)
  {
    using E = EC<C>;
    for (int d = 32; d >= 1; d >>= 1) {
      const typename E::Proj o = proj_shfl_down(v, d);
      if (lane < d) v = E::add(v, o);
    }
    return v;
  }

  template <class C>
  __global__ __launch_bounds__(64) void k_reduce_wave(const typename EC<C>::Proj* __restrict__ buckets, typename EC<C>::Proj* __restrict__ chunkV, typename EC<C>::Proj* __restrict__ chunkT, typename EC<C>::Proj* __restrict__ winsum_direct, uint32_t nb_stride, uint32_t nb_wide, uint32_t m_wide, uint32_t seg_lo, uint32_t nsegr, uint32_t nlo_w, uint32_t nseg_lo, uint32_t m_lo)
  {
    // m_wide / m_lo: rows per lane of a wide / narrow window's chunks (a chunk = 64 m buckets). Round 5 gave both kinds the same m, so
    // a narrow window had half the chunks: 10 x 128 + 2 x 256 = 1792 waves at 2^26 on 1024 SIMDs that hold two of them each -- three
    // quarters of the SIMDs ran two waves back to back, the rest one. With m_lo = m_wide / 2 every window has the same number of
    // chunks (3072 waves of two sizes, dealt out as SIMDs free up). Built, parity-green, measured 3 % slower on the tail: off by default (reduce_group).
    // nb_stride: buckets between the slots of consecutive windows. The first nlo_w windows of the launch are the NARROW
    // windows of a mixed-width plan: they use the first nb_wide / 2 buckets of their slot = nseg_lo chunks; the others use
    // nb_wide buckets, chunks [seg_lo, seg_lo + nsegr). One launch for both kinds: a window set of ~1800 waves fits the chip
    // at once, two launches of the same waves ran one after the other (5.5 against 3.4 ms at 2^26). Chunk outputs of window
    // wp start at wp * nsegr for both kinds (nseg_lo <= nsegr).
    using E = EC<C>;
    const int lane = threadIdx.x;
    const uint32_t nlo_blocks = nlo_w * nseg_lo;
    size_t wp;
    uint32_t ch, nb, lch;
    uint32_t m;
    if (blockIdx.x < nlo_blocks) {
      wp = blockIdx.x / nseg_lo, lch = blockIdx.x % nseg_lo, ch = lch, nb = nb_wide / 2, m = m_lo;
    } else {
      const uint32_t r = blockIdx.x - nlo_blocks;
      wp = nlo_w + r / nsegr, lch = r % nsegr, ch = seg_lo + lch, nb = nb_wide, m = m_wide;
    }
    const size_t oi = wp * nsegr + lch;
    const uint32_t k0 = ch * 64u * m;
    const typename E::Proj* b = buckets + wp * nb_stride;
    typename E::Proj line = E::proj_identity(), tri0 = E::proj_identity();
    for (int i = (int)m - 1; i >= 0; i--) {
      const uint32_t k = k0 + 64u * (uint32_t)i + (uint32_t)lane;
      tri0 = E::add(tri0, line);
      if (k < nb) line = E::add(line, b[k]);
    }
    const typename E::Proj suf = wave_suffix_sum<C>(line, lane); // lane 0: T
    typename E::Proj x = proj_shfl_down(suf, 1);                 // exclusive suffix
    if (lane == 63) x = E::proj_identity();
    for (int q = 0; q < 6; q++)
      tri0 = E::dbl(tri0); // 64 * tri0
    const typename E::Proj v = wave_sum<C>(E::add(x, tri0), lane);
    if (lane == 0) {
      if (winsum_direct) { // the chunk is the whole window (small windows, batches of small MSMs): S = V + T
        winsum_direct[oi] = E::add(v, suf);
      } else {
        chunkV[oi] = v;
        chunkT[oi] = suf;
      }
    }
  }

  // Small windows (nb <= 512 buckets: batches of small MSMs, e.g. the reference's published 2^12 x 2^10 shape with c = 8): a
  // 64-lane chunk per window spends most of its additions in the lane scans -- 128 buckets = 2 rows x 2 adds + 18 scan steps per
  // wave. Here a window gets LW = nb / rows lanes (a power of two, 4 .. 32) and a wave reduces 64 / LW windows at once with the
  // same formula on LW-lane segments: S = T + LW * sum_l tri0_l + sum_l X_l (k0 = 0: the chunk is the whole window).
  // 1024 x 2^12 BN254: bucket reduction 4.47 -> see profiles/r05_notes.md.
  template <class C>
  __global__ __launch_bounds__(64) void k_reduce_small(const typename EC<C>::Proj* __restrict__ buckets, typename EC<C>::Proj* __restrict__ winsum, uint32_t nb, uint32_t lw_log, uint32_t nwindows)
  {
    using E = EC<C>;
    const uint32_t lane = threadIdx.x, LW = 1u << lw_log, rows = nb >> lw_log;
    const uint32_t sl = lane & (LW - 1u);
    const size_t wp = (size_t)blockIdx.x * (64u >> lw_log) + (lane >> lw_log);
    const bool act = wp < nwindows;
    const typename E::Proj* b = buckets + (act ? wp : 0) * nb;
    typename E::Proj line = E::proj_identity(), tri0 = E::proj_identity();
    for (int i = (int)rows - 1; i >= 0; i--) {
      tri0 = E::add(tri0, line);
      if (act) line = E::add(line, b[(uint32_t)i * LW + sl]);
    }
    typename E::Proj suf = line; // inclusive suffix sum over the segment: lane sl gets sum_{sl' >= sl} line
    for (uint32_t d = 1; d < LW; d <<= 1) {
      const typename E::Proj o = proj_shfl_down(suf, (int)d);
      if (sl + d < LW) suf = E::add(suf, o);
    }
    typename E::Proj x = proj_shfl_down(suf, 1); // exclusive
    if (sl == LW - 1u) x = E::proj_identity();
    for (uint32_t q = 0; q < lw_log; q++)
      tri0 = E::dbl(tri0); // LW * tri0
    typename E::Proj v = E::add(x, tri0);
    for (uint32_t d = LW >> 1; d >= 1; d >>= 1) { // segment sum, valid in sl == 0
      const typename E::Proj o = proj_shfl_down(v, (int)d);
      if (sl < d) v = E::add(v, o);
    }
    if (act && sl == 0) winsum[wp] = E::add(v, suf);
  }

  // per window: S = sum_c (V_c + T_c) + chunk * sum_c c * T_c over the nsegr <= blockDim chunks of this device's slice
  // (c = global chunk index = seg_lo + local index)
  template <class C>
  __global__ __launch_bounds__(ReduceWindowLanes<C>::value) void k_reduce_window(const typename EC<C>::Proj* __restrict__ chunkV, const typename EC<C>::Proj* __restrict__ chunkT, typename EC<C>::Proj* __restrict__ winsum, uint32_t nsegr_wide, uint32_t seg_lo_wide, uint32_t log_chunk_wide, uint32_t nlo_w, uint32_t nseg_lo, uint32_t log_chunk_lo)
  {
    // (windows [0, nlo_w) of the launch: narrow windows of a mixed-width plan, nseg_lo chunks of 2^log_chunk_lo buckets from 0; chunk rows are nsegr_wide apart)
    const uint32_t nsegr = blockIdx.x < nlo_w ? nseg_lo : nsegr_wide, seg_lo = blockIdx.x < nlo_w ? 0u : seg_lo_wide;
    const uint32_t log_chunk = blockIdx.x < nlo_w ? log_chunk_lo : log_chunk_wide;
    using E = EC<C>;
    constexpr int RWL = ReduceWindowLanes<C>::value;
    __shared__ typename E::Proj sh[RWL];
    const int NW = blockDim.x / 64; // launched with the power of two >= nsegr (64 .. RWL threads)
    const int wp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool have = (uint32_t)tid < nsegr;
    const typename E::Proj t = have ? chunkT[(size_t)wp * nsegr_wide + tid] : E::proj_identity();
    typename E::Proj u = have ? E::add(chunkV[(size_t)wp * nsegr_wide + tid], t) : E::proj_identity();
    // exclusive suffix sum of T over the block
    const typename E::Proj suf = wave_suffix_sum<C>(t, lane);
    if (lane == 0) sh[wave] = suf; // wave totals
    __syncthreads();
    typename E::Proj x = proj_shfl_down(suf, 1);
    if (lane == 63) x = E::proj_identity();
    typename E::Proj ttot = E::proj_identity(); // sum of T over the whole block (needed for a slice that does not start at 0)
    for (int w = NW - 1; w >= 0; w--) {
      if (w > wave) x = E::add(x, sh[w]);
      ttot = E::add(ttot, sh[w]);
    }
    __syncthreads();
    auto block_sum = [&](typename E::Proj v) { // valid in thread 0
      sh[tid] = v;
      __syncthreads();
      for (int s2 = blockDim.x / 2; s2 >= 1; s2 >>= 1) {
        if (tid < s2) {
          v = E::add(v, sh[tid + s2]);
          sh[tid] = v;
        }
        __syncthreads();
      }
      return v;
    };
    const typename E::Proj U = block_sum(u);
    typename E::Proj X = block_sum(x);
    if (tid == 0) {
      if (seg_lo) X = E::add(X, E::mul_small(ttot, seg_lo));
      for (uint32_t q = 0; q < log_chunk; q++)
        X = E::dbl(X);
      winsum[wp] = E::add(U, X);
    }
  }

  // ---- the same reduction with FOUR lanes per bucket column (round 6): small and mid-size MSMs -------------------------------
  // Up to ~2^20 terms the bucket reduction is a latency chain, not work: at 2^16 its two kernels were half of a 1.0 ms MSM
  // (0.28 + 0.22 ms), one wave per SIMD or less, each lane walking ~50 dependent complete additions of fourteen products. Here a
  // DPP quad owns what a lane owned -- every addition is ec_dbl_quad.hpp's four product rounds, every doubling its two levels --
  // and a wave reduces 16 columns instead of 64. Same sums (S = sum_c (V_c + T_c) + chunk * sum_c c T_c), uniform window plans of a
  // single MSM on one device only; the throughput-bound reductions of the large MSMs keep the one-lane kernels above.
  template <class C>
  __device__ __forceinline__ typename EC<C>::Proj reduce_quad_add(const typename EC<C>::Proj& a, const typename EC<C>::Proj& b, uint32_t role)
  {
    return EcQuadAdd<C>::add(a, b, role);
  }
  template <class C>
  __device__ __forceinline__ typename EC<C>::Proj reduce_quad_dbl(const typename EC<C>::Proj& a, uint32_t role)
  {
    if constexpr (has_small_b3<C>::value)
      return EcDblSmallB<C>::dbl_quad(a, role);
    else
      return EC<C>::dbl(a); // (every lane of the quad computes the same value)
  }
  // chunk ch of window wp = 16 m buckets from ch * 16 m: V = sum (local index) B, T = sum B (lane 0 of the wave stores)
  template <class C>
  __global__ __launch_bounds__(64) void k_reduce_wave_quad(const typename EC<C>::Proj* __restrict__ buckets, typename EC<C>::Proj* __restrict__ chunkV, typename EC<C>::Proj* __restrict__ chunkT, typename EC<C>::Proj* __restrict__ winsum_direct, uint32_t nb, uint32_t m, uint32_t nsegr)
  {
    using E = EC<C>;
    using Proj = typename E::Proj;
    const uint32_t role = threadIdx.x & 3u;
    const int lane = (int)(threadIdx.x >> 2); // 16 quad-lanes per wave
    const size_t wp = blockIdx.x / nsegr;
    const uint32_t ch = blockIdx.x % nsegr;
    const uint32_t k0 = ch * 16u * m;
    const Proj* b = buckets + wp * nb;
    Proj line = E::proj_identity(), tri0 = E::proj_identity();
    for (int i = (int)m - 1; i >= 0; i--) {
      const uint32_t k = k0 + 16u * (uint32_t)i + (uint32_t)lane;
      tri0 = reduce_quad_add<C>(tri0, line, role);
      if (k < nb) line = reduce_quad_add<C>(line, b[k], role); // (the same for the four lanes of a quad)
    }
    Proj suf = line; // inclusive suffix sum over the 16 quad-lanes
    for (int d = 1; d < 16; d <<= 1) {
      const Proj o = proj_shfl_down(suf, 4 * d);
      if (lane + d < 16) suf = reduce_quad_add<C>(suf, o, role);
    }
    Proj x = proj_shfl_down(suf, 4); // exclusive
    if (lane == 15) x = E::proj_identity();
    for (int q = 0; q < 4; q++)
      tri0 = reduce_quad_dbl<C>(tri0, role); // 16 * tri0
    Proj v = reduce_quad_add<C>(x, tri0, role);
    for (int d = 8; d >= 1; d >>= 1) {
      const Proj o = proj_shfl_down(v, 4 * d);
      if (lane < d) v = reduce_quad_add<C>(v, o, role);
    }
    if (lane == 0) {
      const size_t oi = wp * nsegr + ch;
      if (winsum_direct) { // the chunk is the whole window
        const Proj w = reduce_quad_add<C>(v, suf, role);
        if (role == 0) winsum_direct[oi] = w;
      } else if (role == 0) {
        chunkV[oi] = v;
        chunkT[oi] = suf;
      }
    }
  }
  // per window: S = sum_c (V_c + T_c) + chunk * sum_c c T_c over nsegr <= 64 chunks of `chunk` buckets (any size >= 1), one quad per
  // chunk. The two block sums of k_reduce_window are one here: every quad scales its own exclusive suffix by the chunk size first
  // (a double-and-add over the bits of `chunk`, side by side in all quads). Serves the chunks of either wave kernel.
  template <class C>
  __global__ __launch_bounds__(256) void k_reduce_window_quad(const typename EC<C>::Proj* __restrict__ chunkV, const typename EC<C>::Proj* __restrict__ chunkT, typename EC<C>::Proj* __restrict__ winsum, uint32_t nsegr, uint32_t chunk)
  {
    using E = EC<C>;
    using Proj = typename E::Proj;
    __shared__ Proj sh[64];
    const int wp = blockIdx.x, tid = threadIdx.x;
    const uint32_t role = (uint32_t)tid & 3u;
    const int q = tid >> 2, lane = q & 15, wave = q >> 4;
    const int NQ = blockDim.x >> 2, NW = (blockDim.x + 63) / 64; // launched with 4 x (the power of two >= nsegr, at least 16) threads
    const bool have = (uint32_t)q < nsegr;
    const Proj t = have ? chunkT[(size_t)wp * nsegr + q] : E::proj_identity();
    const Proj u = have ? reduce_quad_add<C>(chunkV[(size_t)wp * nsegr + q], t, role) : E::proj_identity();
    Proj suf = t; // inclusive suffix sum of T over the 16 quad-lanes of the wave
    for (int d = 1; d < 16; d <<= 1) {
      const Proj o = proj_shfl_down(suf, 4 * d);
      if (lane + d < 16) suf = reduce_quad_add<C>(suf, o, role);
    }
    if (lane == 0 && role == 0) sh[wave] = suf; // wave totals
    __syncthreads();
    Proj x = proj_shfl_down(suf, 4);
    if (lane == 15) x = E::proj_identity();
    for (int w = NW - 1; w > wave; w--)
      x = reduce_quad_add<C>(x, sh[w], role); // exclusive suffix over the block: sum of T_c' for c' > c
    __syncthreads();
    {
      const Proj x1 = x; // chunk * x, most significant bit first (chunk is the same in every lane)
      for (int bit = 30 - __clz((int)chunk); bit >= 0; bit--) {
        x = reduce_quad_dbl<C>(x, role);
        if ((chunk >> bit) & 1u) x = reduce_quad_add<C>(x, x1, role);
      }
    }
    Proj v = reduce_quad_add<C>(u, x, role);
    if (role == 0) sh[q] = v;
    __syncthreads();
    for (int s2 = NQ / 2; s2 >= 1; s2 >>= 1) {
      if (q < s2) {
        v = reduce_quad_add<C>(v, sh[q + s2], role);
        if (role == 0) sh[q] = v;
      }
      __syncthreads();
    }
    if (tid == 0) winsum[wp] = v;
  }

  // 5c. window combine: result = sum_w 2^(c*w) * winsum[w], written in the reference's
  //     projective_t layout (canonical words). One 128-lane block: lane w scales its own window
  //     sum by c*w doublings (the same critical path as a serial Horner, but the doublings of
  //     different windows overlap; Jacobian doubling chain, 2M + 5S per step), then a tree through
  //     LDS. <1 % of the work at 2^26, but the latency floor of a small MSM.
  //     (A <<<1,1>>> serial Horner is provably wave-uniform, so hipcc compiles ALL of its field
  //     arithmetic to SALU code, which is several times slower per multiply than the VALU path.)
  // k_final: 4 lanes per window, ONE wave (16 windows) per block. A block of several busy waves was measured twice:
  // windows 0..15 in wave 0 and 16.. in wave 1 of one block -> a 17- to 22-window combine takes 0.8-1.0 ms against
  // 0.61-0.73 ms for 13-16 windows of the same chain length; the windows dealt round-robin to four waves -> 1.1-1.6 ms
  // (profiles/r03_notes.md section 8): the waves of one small block share issue bandwidth instead of taking a SIMD each.
  // Separate single-wave blocks land on different CUs; their partial sums are added by k_final_combine.
  constexpr int FINAL_WINDOWS_PER_BLOCK = 16;
  // Block (bw, b) combines windows [w0 + 16 bw, ...) of MSM b, at most 16 of the nw windows [w0, w0 + nw) -- still scaled
  // by their full 2^(c*w). With `partial` the block's sum goes to partial[(slot0 + bw) * nmsm + b] in the kernels' own
  // representation (several blocks per MSM, or a window group of the pipelined schedule); without, to result[b].
  template <class C>
  __global__ __launch_bounds__(64) void k_final(const typename EC<C>::Proj* __restrict__ winsum, uint32_t* __restrict__ result, int wpf, WinWidths ww, int w0, int nw, typename EC<C>::Proj* __restrict__ partial, int slot0, int nmsm)
  {
    const int c = ww.c;
    using E = EC<C>;
    __shared__ typename E::Proj sh[FINAL_WINDOWS_PER_BLOCK];
    // four lanes per window: the doubling chain 2^(c*w) * S_w is the latency floor of the whole MSM, and a quad
    // runs it with three dependent products per step instead of seven (ec.hpp dbl_jac_quad)
    const uint32_t role = threadIdx.x & 3u;
    const int bw = blockIdx.x, b = blockIdx.y;
    const int wfirst = w0 + FINAL_WINDOWS_PER_BLOCK * bw;
    const int nwb = min(FINAL_WINDOWS_PER_BLOCK, w0 + nw - wfirst);
    winsum += (size_t)b * wpf + wfirst;
    const int w = threadIdx.x >> 2;
    typename E::Proj v = E::proj_identity();
    if (w < nwb) {
      v = winsum[w];
      if (wfirst + w > 0) {
        const int wg = wfirst + w; // bit offset of the window = sum of the widths below it
        const int nd = wg < ww.n_lo ? wg * (c - 1) : ww.n_lo * (c - 1) + (wg - ww.n_lo) * c;
        if constexpr (has_small_b3<C>::value) { // complete projective doublings, two product levels per step (ec_dbl_quad.hpp; round 6)
          for (int i = 0; i < nd; i++)          // the same trip count in all four lanes of a quad
            v = EcDblSmallB<C>::dbl_quad(v, role);
        } else { // Jacobian doubling chain (ec.hpp), 2M + 5S in three levels per step
          typename E::Jac j = E::to_jac(v);
          for (int i = 0; i < nd; i++)
            j = E::dbl_jac_quad(j, role);
          v = E::from_jac(j);
        }
      }
    }
    if (role == 0) sh[w] = v;
    __syncthreads();
    const int lane = threadIdx.x;
    for (int s = FINAL_WINDOWS_PER_BLOCK >> 1; s >= 1; s >>= 1) {
      if (lane < s) sh[lane] = E::add(sh[lane], sh[lane + s]);
      __syncthreads();
    }
    if (lane == 0) {
      if (partial)
        partial[(size_t)(slot0 + bw) * nmsm + b] = sh[0];
      else
        E::store_proj_canonical(result + (size_t)b * 3 * E::N32, sh[0]);
    }
  }
  // Window combine of a BATCH: Horner per MSM, one DPP quad each. k_final scales every window by its own doubling chain -- 16 x
  // the doublings of a Horner walk, which is what a single MSM wants (the chains run side by side: the floor is ONE chain) and
  // what a batch does not: 1024 MSMs x 32 windows kept 2048 waves busy for 1.8 ms. With a quad per MSM the batch is 64 waves and
  // ends after one walk of ~250 quad doublings + one quad addition per window.
  template <class C>
  __global__ __launch_bounds__(64) void k_final_horner(const typename EC<C>::Proj* __restrict__ winsum, uint32_t* __restrict__ result, int wpf, WinWidths ww, int nmsm)
  {
    using E = EC<C>;
    const uint32_t role = threadIdx.x & 3u;
    const int b = blockIdx.x * 16 + (int)(threadIdx.x >> 2);
    const bool act = b < nmsm;
    const typename E::Proj* ws = winsum + (size_t)(act ? b : 0) * wpf;
    typename E::Proj acc = ws[wpf - 1];
    for (int w = wpf - 2; w >= 0; w--) { // acc = 2^width(w) * acc + S_w (the same trip counts in every lane)
      const int nd = ww.width(w);
      if constexpr (has_small_b3<C>::value) {
        for (int i = 0; i < nd; i++)
          acc = EcDblSmallB<C>::dbl_quad(acc, role);
        acc = EcQuadAdd<C>::add(acc, ws[w], role);
      } else {
        typename E::Jac j = E::to_jac(acc);
        for (int i = 0; i < nd; i++)
          j = E::dbl_jac_quad(j, role);
        acc = E::add_quad(E::from_jac(j), ws[w], role);
      }
    }
    if (act && role == 0) E::store_proj_canonical(result + (size_t)b * 3 * E::N32, acc);
  }

  // result[b] = sum of the ng group partials of MSM b (partials[g * nmsm + b]), canonical words
  template <class C>
  __global__ __launch_bounds__(64) void k_final_combine(const typename EC<C>::Proj* __restrict__ partials, uint32_t* __restrict__ result, int ng, int nmsm)
  {
    using E = EC<C>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nmsm) return;
    typename E::Proj v = partials[b];
    for (int g = 1; g < ng; g++)
      v = E::add(v, partials[(size_t)g * nmsm + b]);
    E::store_proj_canonical(result + (size_t)b * 3 * E::N32, v);
  }

  // ------------------------------------------------------------------------------------------
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
    typename F::fe zi = F::inv(p.z); // cold path (precompute / generator)
    typename F::fe x = F::mul(p.x, zi), y = F::mul(p.y, zi);
    if (refmont) {
      F::to_refmont(w, x);
      F::to_refmont(w + E::N32, y);
    } else {
      F::to_canonical(w, x);
      F::to_canonical(w + E::N32, y);
    }
  }

  // msm_precompute_bases: out[pf*i + j] = 2^(j*shift) * P_i  (cpu_msm.hpp:455-480), shift = c*wpf.
  // Two kernels, neither with a private array (rounds 3-5 kept up to 32 Jacobian outputs per thread in dynamically indexed arrays:
  // 4.7 - 14.5 KB of scratch per lane, which is what capped the resident waves and left the kernel at 211 ms for 2^24 x 4 against
  // the 135 ms its doublings cost at the issue roof):
  //   k_precompute_chains  one thread per base: the Jacobian doubling chain (dbl-2009-l, 2M + 5S per step, ~254 (pf-1)/pf steps per
  //                        base, ~960 v_mad_u64_u32 each). After every `shift` doublings the point goes to memory: X and Y (reduced
  //                        Montgomery words) into the table slot it will finally occupy, Z into a side buffer. Nothing is kept.
  //   k_precompute_affine  Montgomery's trick over runs of 8 / 32 consecutive table entries: prefix products of the Z's to a
  //                        second side buffer on the way up, ONE inversion per run (rounds 1-3 paid ~314 products per entry), then
  //                        x = X / Z^2, y = Y / Z^3 on the way down, converted to the caller's layout in place.
  // The side buffers hold Z and the prefix for one chunk of bases at a time (msm_precompute_run); their traffic (~5 field elements
  // per entry) is noise beside 254 doublings.
  // Entries per inversion: 8 while the table is small (a run is a serial walk: short runs keep the chip busy), 32 once there are
  // enough runs to fill it -- ~380 products per inversion are 48 per entry at 8 and 12 at 32, against ~590 in the entry's doublings.
  constexpr int PRECOMP_RUN_SMALL = 8, PRECOMP_RUN_LARGE = 32;
  template <class C>
  __device__ __forceinline__ void store_point_words(uint32_t* __restrict__ dst, const uint32_t* w, bool aligned16)
  {
    constexpr int PW = 2 * EC<C>::N32; // 16 / 24 / 32 / 48 words: always a multiple of 4
    if (aligned16) {
#pragma unroll
      for (int k = 0; k < PW; k += 4)
        *reinterpret_cast<uint4*>(dst + k) = make_uint4(w[k], w[k + 1], w[k + 2], w[k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < PW; k++)
        dst[k] = w[k];
    }
  }
  template <class C>
  __device__ __forceinline__ void load_point_words(uint32_t* w, const uint32_t* __restrict__ src, bool aligned16)
  {
    constexpr int PW = 2 * EC<C>::N32;
    if (aligned16) {
#pragma unroll
      for (int k = 0; k < PW; k += 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + k);
        w[k] = v.x, w[k + 1] = v.y, w[k + 2] = v.z, w[k + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PW; k++)
        w[k] = src[k];
    }
  }
  // bases [i0, i0 + cnt) of the table; zbuf is indexed from the chunk's first entry: zbuf[((i - i0) * (pf - 1) + j - 1) * N32]
  template <class C, int MINW>
  __global__ __launch_bounds__(64, MINW) void k_precompute_chains(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ zbuf, long long i0, int cnt, int pf, int shift, bool refmont, bool in16, bool out16)
  {
    using E = EC<C>;
    using F = typename E::F;
    constexpr int PW = 2 * E::N32;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const long long i = i0 + t;
    uint32_t w[PW];
    load_point_words<C>(w, in + (size_t)i * PW, in16);
    store_point_words<C>(out + (size_t)i * pf * PW, w, out16); // entry 0 is the base itself
    typename E::Jac jp;
    bool ident = E::words_are_zero(w); // the identity is (0, 0) in the reference's affine layout
    if (!ident) {
      jp.x = refmont ? F::from_refmont(w) : F::from_canonical(w);
      jp.y = refmont ? F::from_refmont(w + E::N32) : F::from_canonical(w + E::N32);
      jp.z = F::one();
    }
    for (int j = 1; j < pf; j++) {
      if (!ident) {
        for (int sft = 0; sft < shift; sft++)
          jp = E::dbl_jac_lazy(jp); // (G1: two conditional subtractions per step instead of eight, ec.hpp)
        ident = F::is_zero(jp.z); // a point of order two doubles to Z = 0 (and stays there)
      }
      uint32_t* zdst = zbuf + ((size_t)t * (pf - 1) + (j - 1)) * E::N32;
      if (ident) { // Z = 0 marks the entry for k_precompute_affine
#pragma unroll
        for (int k = 0; k < E::N32; k++)
          zdst[k] = 0;
        continue;
      }
      F::pack(w, F::reduce(jp.x));
      F::pack(w + E::N32, F::reduce(jp.y));
      store_point_words<C>(out + ((size_t)i * pf + j) * PW, w, out16);
      uint32_t zw[E::N32];
      F::pack(zw, F::reduce(jp.z));
#pragma unroll
      for (int k = 0; k < E::N32; k++)
        zdst[k] = zw[k];
    }
  }
  // entries q of the chunk (q = (i - i0) * (pf - 1) + j - 1, q < nq): one thread per run of `run` consecutive q
  template <class C, int MINW>
  __global__ __launch_bounds__(64, MINW) void k_precompute_affine(uint32_t* __restrict__ out, const uint32_t* __restrict__ zbuf, uint32_t* __restrict__ pbuf, long long i0, long long nq, int pf, bool refmont, bool out16, int run)
  {
    using E = EC<C>;
    using F = typename E::F;
    using fe = typename F::fe;
    constexpr int PW = 2 * E::N32;
    const long long q0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * run;
    if (q0 >= nq) return;
    const int m = (int)(nq - q0 < run ? nq - q0 : run);
    fe acc = F::one();
    for (int k = 0; k < m; k++) { // prefix products (of the non-zero Z's) before entry k
      uint32_t zw[E::N32];
#pragma unroll
      for (int e = 0; e < E::N32; e++)
        zw[e] = zbuf[(size_t)(q0 + k) * E::N32 + e];
      uint32_t pw[E::N32];
      F::pack(pw, F::reduce(acc));
#pragma unroll
      for (int e = 0; e < E::N32; e++)
        pbuf[(size_t)(q0 + k) * E::N32 + e] = pw[e];
      uint32_t any = 0;
#pragma unroll
      for (int e = 0; e < E::N32; e++)
        any |= zw[e];
      if (any) acc = F::mul(acc, F::unpack(zw));
    }
    fe inv = F::inv(acc);
    for (int k = m - 1; k >= 0; k--) {
      const long long q = q0 + k;
      const long long slot = (i0 + q / (pf - 1)) * pf + 1 + q % (pf - 1);
      uint32_t* dst = out + (size_t)slot * PW;
      uint32_t zw[E::N32], w[PW];
#pragma unroll
      for (int e = 0; e < E::N32; e++)
        zw[e] = zbuf[(size_t)q * E::N32 + e];
      uint32_t any = 0;
#pragma unroll
      for (int e = 0; e < E::N32; e++)
        any |= zw[e];
      if (!any) {
#pragma unroll
        for (int e = 0; e < PW; e++)
          w[e] = 0;
        store_point_words<C>(dst, w, out16);
        continue;
      }
      uint32_t pw[E::N32];
#pragma unroll
      for (int e = 0; e < E::N32; e++)
        pw[e] = pbuf[(size_t)q * E::N32 + e];
      const fe zi = F::mul(inv, F::unpack(pw)); // 1 / Z_k
      inv = F::mul(inv, F::unpack(zw));
      const fe zi2 = F::sqr(zi);
      load_point_words<C>(w, dst, out16);
      const fe x = F::mul(F::unpack(w), zi2), y = F::mul(F::unpack(w + E::N32), F::mul(zi2, zi));
      if (refmont) {
        F::to_refmont(w, x);
        F::to_refmont(w + E::N32, y);
      } else {
        F::to_canonical(w, x);
        F::to_canonical(w + E::N32, y);
      }
      store_point_words<C>(dst, w, out16);
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
  // Block b sums points pts[b*3*N32 + i*stride], i < n (stride in words; gridDim.x = batch: the partials of batch
  // element b sit `stride` words apart) into out[b*3*N32].
  template <class C>
  __global__ __launch_bounds__(64) void k_proj_sum(const uint32_t* __restrict__ pts, int n, size_t stride, uint32_t* __restrict__ out)
  {
    using E = EC<C>;
    using F = typename E::F;
    __shared__ typename E::Proj sh[64];
    const int lane = threadIdx.x;
    pts += (size_t)blockIdx.x * 3 * E::N32;
    out += (size_t)blockIdx.x * 3 * E::N32;
    typename E::Proj v = E::proj_identity();
    for (int i = lane; i < n; i += 64) {
      const uint32_t* w = pts + (size_t)i * stride;
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
    k_proj_sum<C><<<1, 64, 0, st>>>((const uint32_t*)pts, n, (size_t)3 * EC<C>::N32, (uint32_t*)out);
    LAUNCH_CHECK("k_proj_sum", st);
    return ICICLE_SUCCESS;
  }

  // dst[i] += src[i] over bucket arrays (multi-device bucket exchange: partial bucket sums of shards / peers)
  template <class C>
  __global__ __launch_bounds__(128) void k_bucket_add(typename EC<C>::Proj* __restrict__ dst, const typename EC<C>::Proj* __restrict__ src, size_t n)
  {
    using E = EC<C>;
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    dst[t] = E::add(dst[t], src[t]);
  }

  // Hook between bucket accumulation and bucket reduction (multi-device variant E2, msm_multi.hpp): may replace the
  // bucket contents with sums over shards / devices, skip this call's reduction, or restrict it to a segment range.
  template <class C>
  struct MsmBucketHook {
    virtual ~MsmBucketHook() = default;
    virtual icicle_error_t after_accumulate(typename EC<C>::Proj* buckets, size_t tw, uint32_t nb, uint32_t nseg, uint32_t m, hipStream_t st, bool* skip_reduce, uint32_t* seg_lo, uint32_t* nsegr) = 0;
  };

  // ------------------------------------------------------------------------------------------
  template <class C>
  static icicle_error_t msm_run_single_planned(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v, MsmBucketHook<C>* hook, int plan_override);
  // The auto-selected mixed-width plan (2^23 terms and up) holds ~2.15 x the buckets, counters and lists of the uniform plan -- several
  // GB more for G2 at 2^26. A call that does not get that memory is run again on the uniform plan, which fit before round 5.
  template <class C>
  static icicle_error_t msm_run_single(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v, MsmBucketHook<C>* hook = nullptr)
  {
    // (icicle_hip_test_inject_failure(0, 9): the first attempt "runs out of memory" once -- the rehearsal of this path)
    icicle_error_t rc = (!hook && test_failure_armed(0, 9)) ? ICICLE_ALLOCATION_FAILED : msm_run_single_planned<C>(scalars_v, bases_v, n, cfg, results_v, hook, 0);
    if (rc == ICICLE_ALLOCATION_FAILED && !hook && cfg && make_plan(std::max(n, 1), C::fr::NBITS, *cfg, 0).n_lo > 0) {
      multi_stats().plan_fallbacks++;
      (void)hipGetLastError();
      if (cfg->stream) (void)hipStreamSynchronize((hipStream_t)cfg->stream);
      (void)icicle_hip_release_workspace();
      rc = msm_run_single_planned<C>(scalars_v, bases_v, n, cfg, results_v, hook, -1);
    }
    return rc;
  }
  template <class C>
  static icicle_error_t msm_run_single_planned(const void* scalars_v, const void* bases_v, int n, const icicle_msm_config_t* cfg, void* results_v, MsmBucketHook<C>* hook, int plan_override)
  {
    using E = EC<C>;
    using FR = FieldOps<typename C::fr>;
    constexpr int PW = 2 * E::N32, RW = 3 * E::N32;
    if (!cfg || !results_v || n < 0) return ICICLE_INVALID_ARGUMENT;
    const int batch = std::max(1, cfg->batch_size);
    if (n > 0 && (!scalars_v || !bases_v)) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    int force_windows = hook ? -1 : plan_override; // (a bucket exchange adds bucket arrays of several calls element-wise: uniform widths)
    if (!hook && plan_override == 0 && cfg->ext) force_windows = std::max(0, reinterpret_cast<const ConfigExt*>(cfg->ext)->get_int("hip_msm_windows", 0));
    {
      static const int env_w = getenv("ICICLE_HIP_MSM_WINDOWS") ? atoi(getenv("ICICLE_HIP_MSM_WINDOWS")) : 0; // A/B: -1 = uniform only
      if (force_windows == 0 && env_w != 0) force_windows = env_w;
    }
    const MsmPlan pl = make_plan(std::max(n, 1), C::fr::NBITS, *cfg, force_windows);
    const WinWidths ww{pl.c, pl.n_lo, pl.negate};
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

    // ---- geometry
    const uint32_t nb = pl.nb;
    const int wpf = pl.wpf;
    const size_t cap = npts_one; // bucket-list capacity per target window
    SortPlan sp;
    {
      const int kb = pl.c - 1;
      // each tile-sorted pass ranks into <= 1024 destinations (one per thread of the block); small
      // windows (kb <= 10) need a single level: pass A already produces the bucket lists
      int hb = kb <= 10 ? kb : (kb + 1) / 2;
      if (const char* e = getenv("ICICLE_HIP_MSM_HB")) hb = atoi(e);
      hb = std::max(kb - 11, std::min(std::min(kb, 10), hb)); // pass A: at most 2^10 partitions; pass B: up to 2^11 bins (c = 22)
      if (hb < 0 || hb > 10 || kb - hb > 11) return ICICLE_INVALID_ARGUMENT;
      sp.hb = hb;
      sp.lb = kb - hb;
      sp.jb = 0;
      while ((1 << sp.jb) < pf)
        sp.jb++;
      int logn = 0;
      while (((size_t)1 << logn) < (size_t)n)
        logn++;
      const int max_chunk = 31 - sp.lb - sp.jb;
      if (max_chunk < 10) return ICICLE_INVALID_ARGUMENT;
      sp.chunk_log = std::max(10, std::min(max_chunk, std::max(17, logn - 9)));
      if (sp.chunk_log < logn - 9) return ICICLE_INVALID_ARGUMENT; // would need more than 512 pass-A blocks per window
      sp.nblk = (int)(((size_t)n + ((size_t)1 << sp.chunk_log) - 1) >> sp.chunk_log);
    }
    const bool single_level = (sp.lb == 0);
    const size_t nparts_w = (size_t)1 << sp.hb;
    // bucket reduction geometry: a wave owns a chunk of 64*mrow buckets; at most RWL chunks per window (one thread of
    // the per-window kernel each), at least 16 rows per lane when the window is big enough to amortise the wave scans
    constexpr uint32_t RWL = ReduceWindowLanes<C>::value;
    // Rows per lane: 16 amortise the wave scans; a single mid-size MSM has fewer reduction waves than the chip has SIMDs
    // and is bound by the length of one wave's chain, so there the rows shrink until the waves fill the SIMDs once
    // (2^16, c = 15: 272 waves of 16 rows -> 544 of 8, 1.78 -> 1.59 ms; at 2^20 16 rows already make 960 waves). The rule
    // depends on the window plan only (every shard of a multi-device call derives the same geometry).
    uint32_t mrow_cap = 16;
    if (batch == 1 && nb >= 8192) {
      uint32_t need = (uint32_t)(((uint64_t)pl.wpf * nb + 65535) / 65536), p2 = 2;
      while (p2 < need)
        p2 <<= 1;
      mrow_cap = std::min<uint32_t>(16, p2);
    }
    if (const char* e = getenv("ICICLE_HIP_MSM_MROW")) mrow_cap = (uint32_t)std::max(1, atoi(e)); // (A/B knob)
    const uint32_t mrow = std::max<uint32_t>(1, std::max<uint32_t>(nb / (64 * RWL), std::min<uint32_t>(mrow_cap, nb / 64)));
    const uint32_t m = 64 * mrow;                       // buckets per chunk ("segment" of the exchange hook)
    const uint32_t nseg = std::max<uint32_t>(1, nb / m); // chunks per window
    uint32_t log_chunk = 0;
    while ((1u << log_chunk) < m)
      log_chunk++;

    // ---- batch folding: BB MSMs of the batch run as ONE launch sequence with BB*wpf windows
    // (wrappers/rust/icicle-core/src/msm/tests.rs:92-254 batches; small MSMs would otherwise leave the
    // GPU idle). BB is bounded by a memory budget and by the grid.y limit.
    const size_t per_msm_bytes = (size_t)pl.nwin * n * 4 + (single_level ? 1 : 2) * (size_t)wpf * cap * 4 + (size_t)wpf * nb * (sizeof(typename E::Proj) + 12) +
                                 (size_t)wpf * nparts_w * sp.nblk * 8 + (shared || !(cfg->are_points_montgomery_form) ? 0 : npts_one * PW * 4);
    size_t budget = (size_t)48 << 30;
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 2);
    }
    int BB = (int)std::max<size_t>(1, std::min<size_t>((size_t)batch, budget / std::max<size_t>(per_msm_bytes, 1)));
    BB = std::min(BB, std::max(1, 60000 / wpf));
    if (hook && BB < batch) { // a bucket exchange needs the whole batch in one launch group (msm_multi_run checked the grid limit)
      BB = batch;
      if ((size_t)BB * wpf > 60000) return ICICLE_INVALID_ARGUMENT;
    }
    const size_t TW = (size_t)BB * wpf; // windows per launch
    const size_t nbk = TW * nb;
    const size_t nparts = TW << sp.hb;
    const size_t tabA = nparts * sp.nblk + TW; // [wl][h][b] counters + per-window totals
    const size_t elems_max = (size_t)BB * n * pl.nwin + 1;
    const uint32_t maxblkB = (uint32_t)std::min<size_t>(nparts + (elems_max >> CHUNKB_LOG) + 2, 0x7fffffffu);
    const uint32_t ovf_cap = (uint32_t)std::min<size_t>(elems_max / pl.seg + 16, 0x7fffffffu);

    TempBuf d_mont, d_dig, d_partA, d_sorted, d_cntA, d_offA, d_bstart, d_count, d_offs, d_cursor, d_buckets, d_seg, d_win, d_ovf, d_ovfpart, d_ovfcnt, d_scansum, d_perm, d_sztab, d_szoff, d_firsts;
    // bases are gathered where they lie unless they need converting (reference-Montgomery input) or realigning
    const bool stage_bases = cfg->are_points_montgomery_form || (((uintptr_t)d_bases) & 15) != 0;
    if (stage_bases) HIP_TRY(d_mont.alloc((shared ? 1 : (size_t)BB) * npts_one * PW * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_dig.alloc((size_t)BB * pl.nwin * n * 4, st), ICICLE_ALLOCATION_FAILED);
    if (!single_level) HIP_TRY(d_partA.alloc(TW * cap * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_sorted.alloc(TW * cap * 4 + 64, st), ICICLE_ALLOCATION_FAILED); // + slack: k_accumulate reads lists in groups of 4
    HIP_TRY(d_cntA.alloc(tabA * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_offA.alloc(tabA * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_bstart.alloc((nparts + 1 + 16) * 4, st), ICICLE_ALLOCATION_FAILED); // + one end marker per window group
    HIP_TRY(d_count.alloc(nbk * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_offs.alloc(nbk * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_cursor.alloc(nbk * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_buckets.alloc(nbk * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_seg.alloc(2 * TW * std::max<size_t>(nseg, 64) * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED); // chunk V | chunk T (64: the quad reduction's chunks per window)
    HIP_TRY(d_win.alloc(TW * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_ovf.alloc(((size_t)ovf_cap + 16 * 16) * sizeof(OvfSeg), st), ICICLE_ALLOCATION_FAILED); // (+ the per-group slack of the pipelined schedule)
    HIP_TRY(d_ovfpart.alloc(((size_t)ovf_cap + 16 * 16) * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_firsts.alloc(((size_t)ovf_cap + 16 * 17) * 4 + 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_ovfcnt.alloc(16 * 16, st), ICICLE_ALLOCATION_FAILED); // [overflow segments, overflowing buckets, largest bucket, -] per window group
    const size_t szblk_max = (nbk + SZ_CHUNK - 1) / SZ_CHUNK + 16; // (+ one partial block per window group)
    HIP_TRY(d_perm.alloc(nbk * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_sztab.alloc(szblk_max * SZ_BINS * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_szoff.alloc(szblk_max * SZ_BINS * 4, st), ICICLE_ALLOCATION_FAILED);
    HIP_TRY(d_scansum.alloc(TW * (std::max<size_t>((size_t)sp.nblk << sp.hb, nb) / SCAN_CHUNK + 1) * 4, st), ICICLE_ALLOCATION_FAILED);

    const size_t ldsA = tile_lds_bytes(1u << sp.hb) + ((size_t)4 << sp.hb);
    const size_t ldsB = tile_lds_bytes(1u << sp.lb) + ((size_t)sp.nblk + 1) * 4 + (SORT_TS / 64) * 4;
    constexpr int CO_TPB = 256; // block size of the co-resident sort (window groups, below)
    constexpr uint32_t CO_TS = CO_TPB * SORT_EPT;
    const size_t ldsA_co = tile_lds_bytes(1u << sp.hb, CO_TS) + ((size_t)4 << sp.hb);
    const size_t ldsB_co = tile_lds_bytes(1u << sp.lb, CO_TS) + ((size_t)sp.nblk + 1) * 4 + (CO_TS / 64) * 4;
    if (ldsA > 156 * 1024 || ldsB > 156 * 1024) return ICICLE_INVALID_ARGUMENT;
    HIP_TRY(hipFuncSetAttribute((const void*)k_a_scatter<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipFuncSetAttribute((const void*)k_a_scatter<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipFuncSetAttribute((const void*)k_b_count, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024), ICICLE_INVALID_ARGUMENT);
    HIP_TRY(hipFuncSetAttribute((const void*)k_b_scatter<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024), ICICLE_INVALID_ARGUMENT);

    uint32_t* dig = d_dig.as<uint32_t>();
    uint32_t* cntA = d_cntA.as<uint32_t>();
    uint32_t* offA = d_offA.as<uint32_t>();
    uint32_t* sorted = d_sorted.as<uint32_t>();
    uint32_t* count = d_count.as<uint32_t>();
    uint32_t* offs = d_offs.as<uint32_t>();
    typename E::Proj* buckets = d_buckets.as<typename E::Proj>();

    for (int b0 = 0; b0 < batch; b0 += BB) {
      const int bb = std::min(BB, batch - b0); // MSMs in this launch sequence
      const size_t tw = (size_t)bb * wpf;
      const size_t gbk = tw * nb;
      const size_t gparts = tw << sp.hb;
      const uint32_t* acc_bases = d_bases + (shared ? 0 : (size_t)b0 * npts_one * PW);
      if (stage_bases) {
        if (b0 == 0 || !shared) {
          const size_t ncoord = (shared ? 1 : (size_t)bb) * npts_one * 2 * (E::N32 / FieldOps<typename C::fq>::N32);
          k_bases_stage<C><<<dim3((unsigned)((ncoord + 255) / 256)), 256, 0, st>>>(acc_bases, d_mont.as<uint32_t>(), ncoord, cfg->are_points_montgomery_form);
          LAUNCH_CHECK("k_bases_stage", st);
        }
        acc_bases = d_mont.as<uint32_t>();
      }
      KernelTimer::begin(2, st);
      const uint32_t* sc = d_scalars + (size_t)b0 * n * FR::N32;
      const size_t nscal = (size_t)bb * n;
      const size_t lds_dc = ((size_t)wpf << sp.hb) * 4;
      // digits + pass-A histogram in one pass over the scalars, when there are enough chunks to fill the chip
      // (a block walks a whole chunk; with few chunks the thread-per-scalar k_digits + k_a_count pair is faster)
      if (lds_dc <= 64 * 1024 && bb <= 65535 && (size_t)sp.nblk * bb >= 512) {
        k_digits_count<C><<<dim3(sp.nblk, (unsigned)bb), 1024, lds_dc, st>>>(sc, dig, cntA, n, ww, pl.nwin, wpf, sp, cfg->are_scalars_montgomery_form, pl.bits);
        LAUNCH_CHECK("k_digits_count", st);
      } else {
        k_digits<C><<<(unsigned)((nscal + 255) / 256), 256, 0, st>>>(sc, dig, n, nscal, ww, pl.nwin, cfg->are_scalars_montgomery_form, pl.bits);
        LAUNCH_CHECK("k_digits", st);
        k_a_count<<<dim3(sp.nblk, (unsigned)tw), 1024, ((size_t)4 << sp.hb), st>>>(dig, cntA, n, pl.nwin, wpf, pf, sp);
        LAUNCH_CHECK("k_a_count", st);
      }
      // the per-window totals live right after the [tw][2^hb][nblk] table of THIS launch
      {
        const uint32_t m = (uint32_t)(nparts_w * sp.nblk), nch = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
        k_scan_sums<<<dim3(nch, (unsigned)tw), 1024, 0, st>>>(cntA, d_scansum.as<uint32_t>(), m);
        k_scan_apply<<<dim3(nch, (unsigned)tw), 1024, 0, st>>>(cntA, d_scansum.as<uint32_t>(), offA, nullptr, offA + (size_t)tw * m, m);
      }
      LAUNCH_CHECK("k_scan_a", st);

      // ---- window groups ("pipelined schedule"). Everything behind the digit pass is independent
      // per window, and the phases are bound by different things: the sort by memory latency, bucket accumulation by
      // VALU issue, the window combine by the latency of one doubling chain. A single large MSM is therefore cut into
      // NG groups of windows, HIGHEST windows first: while group g is accumulated on the main stream, group g + 1 is
      // sorted on a second stream and group g - 1 is reduced and scaled by its 2^(c*w) on a third, so the long doubling
      // chains of the high windows and most of the sort leave the critical path. NG = 1 is the plain sequence.
      // Batches (bb > 1) already fill the chip with independent work, and a bucket-exchange hook wants all windows at once.
      int glo[MSM_MAX_GROUPS], ghi[MSM_MAX_GROUPS];
      int want_groups = 1;
      if (bb == 1 && !hook && tw >= 4) {
        static const int env_ng = getenv("ICICLE_HIP_MSM_GROUPS") ? atoi(getenv("ICICLE_HIP_MSM_GROUPS")) : MSM_DEFAULT_GROUPS;
        want_groups = env_ng;
      }
      const int NG = msm_window_groups((int)tw, want_groups, glo, ghi); // msm_plan.h
      hipStream_t s_sort = st, s_red = st, s_acc = st;
      if (NG > 1) {
        s_sort = side_stream(10);
        s_red = side_stream(11);
        if (!s_sort || !s_red) return ICICLE_STREAM_CREATION_FAILED;
      }
      // scratch that concurrent groups must not share
      TempBuf d_gscan, d_part;
      const size_t scan_words = tw * (std::max<size_t>((size_t)sp.nblk << sp.hb, nb) / SCAN_CHUNK + 1) + (szblk_max * SZ_BINS) / SCAN_CHUNK + 16;
      int part_slot[MSM_MAX_GROUPS], part_slots = 0; // partial-sum slots of the window combine: one per 16 windows of a group
      for (int g = 0; g < NG; g++) {
        part_slot[g] = part_slots;
        part_slots += (ghi[g] - glo[g] + FINAL_WINDOWS_PER_BLOCK - 1) / FINAL_WINDOWS_PER_BLOCK;
      }
      // (one group: blocks of 16 windows per MSM of the launch; several groups: bb == 1 and the slots counted above)
      HIP_TRY(d_part.alloc((size_t)(NG > 1 ? part_slots : (wpf + FINAL_WINDOWS_PER_BLOCK - 1) / FINAL_WINDOWS_PER_BLOCK) * bb * sizeof(typename E::Proj), st), ICICLE_ALLOCATION_FAILED);
      if (NG > 1) {
        HIP_TRY(d_gscan.alloc((size_t)NG * scan_words * 4, st), ICICLE_ALLOCATION_FAILED);
        HIP_TRY(hipMemsetAsync(d_ovfcnt.ptr(), 0, 16 * 16, st), ICICLE_COPY_FAILED);
      }
      hipEvent_t ev_ready = nullptr, ev_sorted[16] = {nullptr}, ev_acc[16] = {nullptr}, ev_red[16] = {nullptr};
      if (NG > 1) {
        ev_ready = ring_event();
        if (!ev_ready) return ICICLE_ALLOCATION_FAILED;
        HIP_TRY(hipEventRecord(ev_ready, st), ICICLE_SYNCHRONIZATION_FAILED); // digits, pass-A tables and every buffer lease are in place
        HIP_TRY(hipStreamWaitEvent(s_sort, ev_ready, 0), ICICLE_SYNCHRONIZATION_FAILED);
      }
      typename E::Proj* chunkV = d_seg.as<typename E::Proj>();
      typename E::Proj* chunkT = chunkV + TW * std::max<size_t>(nseg, 64);
      uint32_t ovf_off = 0;
      uint32_t g_ovf_off[16], g_ovf_cap[16];
      for (int g = 0; g < NG; g++) { // overflow segments of a group <= (list entries of its windows) / seg; the caps add up to ovf_cap + 16 NG
        size_t src = 0; // source windows folded into the group's target windows (precompute: j * wpf + wp < nwin)
        for (int wp = glo[g]; wp < ghi[g]; wp++)
          for (int j = 0; j < pf; j++)
            src += (j * wpf + wp % wpf < pl.nwin) ? 1 : 0;
        g_ovf_off[g] = ovf_off;
        g_ovf_cap[g] = NG == 1 ? ovf_cap : (uint32_t)std::min<size_t>(((size_t)n * src) / pl.seg + 16, ovf_cap);
        ovf_off += g_ovf_cap[g];
      }

      // ---- phase 1 of a group: bucket lists (pass-A scatter, pass B, overflow plan, size-balanced order)
      // groups behind the first are sorted WHILE the group before them is accumulated: small blocks that fit beside the
      // padded k_accumulate (two waves per SIMD). Only the 9-limb G1 curves leave the registers for that.
      constexpr bool CAN_CORESIDE = !(sizeof(typename E::XYZZ) > 256) && E::F::N <= 9;
      static const bool co_env = !(getenv("ICICLE_HIP_MSM_CORESIDENT") && atoi(getenv("ICICLE_HIP_MSM_CORESIDENT")) == 0);
      const bool co_ok = CAN_CORESIDE && co_env && NG > 1 && ldsA_co <= 60 * 1024 && ldsB_co <= 60 * 1024;
      auto sort_group = [&](int g, hipStream_t sq) -> icicle_error_t {
        const bool coresident = co_ok && g > 0;
        const int w0 = glo[g], nw = ghi[g] - glo[g];
        const size_t bk0 = (size_t)w0 * nb, nbk_g = (size_t)nw * nb;
        const uint32_t p0 = (uint32_t)((size_t)w0 << sp.hb), np_g = (uint32_t)((size_t)nw << sp.hb);
        uint32_t* scan_g = NG > 1 ? d_gscan.as<uint32_t>() + (size_t)g * scan_words : d_scansum.as<uint32_t>();
        if (single_level) {
          if (coresident) k_a_scatter<true, CO_TPB><<<dim3(sp.nblk, (unsigned)nw), CO_TPB, ldsA_co, sq>>>(dig, offA, sorted, n, pl.nwin, wpf, pf, sp, cap, w0);
          else k_a_scatter<true, 1024><<<dim3(sp.nblk, (unsigned)nw), 1024, ldsA, sq>>>(dig, offA, sorted, n, pl.nwin, wpf, pf, sp, cap, w0);
          LAUNCH_CHECK("k_a_scatter<final>", sq);
          k_tables_from_a<<<(unsigned)((nbk_g + 255) / 256), 256, 0, sq>>>(offA, count, offs, nbk_g, nb, sp.nblk, gparts * sp.nblk, bk0);
          LAUNCH_CHECK("k_tables_from_a", sq);
        } else {
          uint32_t* partA = d_partA.as<uint32_t>();
          uint32_t* bstart = d_bstart.as<uint32_t>() + p0 + (NG - 1 - g); // nparts_g + 1 entries per group, groups in window order
          const size_t elems = (size_t)bb * n * pf * (size_t)nw + 1;
          const uint32_t nblkB = (uint32_t)std::min<size_t>(np_g + (elems >> CHUNKB_LOG) + 2, maxblkB);
          if (coresident) k_a_scatter<false, CO_TPB><<<dim3(sp.nblk, (unsigned)nw), CO_TPB, ldsA_co, sq>>>(dig, offA, partA, n, pl.nwin, wpf, pf, sp, cap, w0);
          else k_a_scatter<false, 1024><<<dim3(sp.nblk, (unsigned)nw), 1024, ldsA, sq>>>(dig, offA, partA, n, pl.nwin, wpf, pf, sp, cap, w0);
          LAUNCH_CHECK("k_a_scatter", sq);
          k_b_plan<<<1, 1024, 0, sq>>>(offA, bstart, np_g, (int)tw, sp.hb, sp.nblk, p0);
          LAUNCH_CHECK("k_b_plan", sq);
          HIP_TRY(hipMemsetAsync(count + bk0, 0, nbk_g * 4, sq), ICICLE_COPY_FAILED);
          k_b_count<<<nblkB, 1024, ((size_t)1 << sp.lb) * 4, sq>>>(partA, offA, bstart, count, np_g, (int)tw, sp, cap, nb, p0);
          LAUNCH_CHECK("k_b_count", sq);
          {
            const uint32_t nch = (nb + SCAN_CHUNK - 1) / SCAN_CHUNK;
            k_scan_sums<<<dim3(nch, (unsigned)nw), 1024, 0, sq>>>(count + bk0, scan_g, nb);
            k_scan_apply<<<dim3(nch, (unsigned)nw), 1024, 0, sq>>>(count + bk0, scan_g, offs + bk0, d_cursor.as<uint32_t>() + bk0, nullptr, nb);
          }
          LAUNCH_CHECK("k_scan_buckets", sq);
          if (coresident) k_b_scatter<CO_TPB><<<nblkB, CO_TPB, ldsB_co, sq>>>(partA, offA, bstart, d_cursor.as<uint32_t>(), sorted, np_g, (int)tw, pf, sp, cap, nb, p0);
          else k_b_scatter<1024><<<nblkB, 1024, ldsB, sq>>>(partA, offA, bstart, d_cursor.as<uint32_t>(), sorted, np_g, (int)tw, pf, sp, cap, nb, p0);
          LAUNCH_CHECK("k_b_scatter", sq);
        }
        uint32_t* ovfcnt = d_ovfcnt.as<uint32_t>() + 4 * g;
        if (NG == 1) HIP_TRY(hipMemsetAsync(ovfcnt, 0, 16, sq), ICICLE_COPY_FAILED);
        k_plan_overflow<<<(unsigned)((nbk_g + 255) / 256), 256, 0, sq>>>(count, nbk_g, pl.seg, ovfcnt, d_ovf.as<OvfSeg>() + g_ovf_off[g], d_firsts.as<uint32_t>() + g_ovf_off[g] + g, g_ovf_cap[g], bk0);
        LAUNCH_CHECK("k_plan_overflow", sq);
        {
          const unsigned szblk = (unsigned)((nbk_g + SZ_CHUNK - 1) / SZ_CHUNK);
          const size_t sz0 = (bk0 / SZ_CHUNK + (size_t)(NG - 1 - g)) * SZ_BINS; // this group's slice of the size tables (groups in window order)
          const uint32_t m = szblk * SZ_BINS, nch = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
          uint32_t* sztab = d_sztab.as<uint32_t>() + sz0;
          uint32_t* szoff = d_szoff.as<uint32_t>() + sz0;
          k_bsize_count<<<szblk, 1024, 0, sq>>>(count, sztab, nbk_g, pl.seg, bk0, nb, (uint32_t)pl.n_lo);
          k_scan_sums<<<dim3(nch, 1), 1024, 0, sq>>>(sztab, scan_g, m);
          k_scan_apply<<<dim3(nch, 1), 1024, 0, sq>>>(sztab, scan_g, szoff, nullptr, nullptr, m);
          k_bsize_scatter<<<szblk, 1024, 0, sq>>>(count, szoff, d_perm.as<uint32_t>() + bk0, nbk_g, pl.seg, bk0, nb, (uint32_t)pl.n_lo);
          LAUNCH_CHECK("k_bsize_scatter", sq);
        }
        return ICICLE_SUCCESS;
      };

      // ---- phase 2 of a group: bucket accumulation (+ the fold of the overflow partials)
      auto accumulate_group = [&](int g, hipStream_t sq) -> icicle_error_t {
        const int w0 = glo[g], nw = ghi[g] - glo[g];
        const size_t bk0 = (size_t)w0 * nb, nbk_g = (size_t)nw * nb;
        uint32_t* ovfcnt = d_ovfcnt.as<uint32_t>() + 4 * g;
        OvfSeg* ovf = d_ovf.as<OvfSeg>() + g_ovf_off[g];
        typename E::Proj* ovfpart = d_ovfpart.as<typename E::Proj>() + g_ovf_off[g];
        const uint32_t ocap = g_ovf_cap[g];
        KernelTimer::begin(0, sq);
        {
          // waves per SIMD the register allocator must leave room for: 3 fits BN254 G1 (157 VGPRs) without
          // spilling, 2 fits BLS12-381 G1 (14-limb elements) and BN254 G2, 1 for BLS12-381 G2
          constexpr bool BIGPT = sizeof(typename E::XYZZ) > 256; // G2
          static const int minw = getenv("ICICLE_HIP_MSM_ACC_WAVES") ? atoi(getenv("ICICLE_HIP_MSM_ACC_WAVES"))
                                                                     : (BIGPT ? (sizeof(typename E::XYZZ) <= 288 ? 2 : 1) : (E::F::N <= 9 ? 3 : 2));
          // threads = the group's buckets in use (perm[] lists exactly those) + its overflow slots
          const int nlo_g = std::max(0, std::min(pl.n_lo, w0 + nw) - w0); // (bb == 1 whenever n_lo > 0)
          const size_t nbk_used = nbk_g - (size_t)nlo_g * (nb / 2);
          const size_t nthreads_acc = nbk_used + ocap;
          const unsigned gridn = (unsigned)((nthreads_acc + 127) / 128);
          const size_t bstride = shared ? 0 : npts_one * PW;
#define ACC_ARGS acc_bases, sorted, count, offs, buckets, ovfpart, ovf, ovfcnt, d_perm.as<uint32_t>() + bk0, ocap, nb, nbk_used, cap, pl.seg, wpf, bstride
          if constexpr (BIGPT) {
            if constexpr (sizeof(typename E::XYZZ) <= 288) {
              if (minw >= 2) k_accumulate<C, 2><<<gridn, 128, 0, sq>>>(ACC_ARGS);
              else k_accumulate<C, 1><<<gridn, 128, 0, sq>>>(ACC_ARGS);
            } else {
              k_accumulate<C, 1><<<gridn, 128, 0, sq>>>(ACC_ARGS);
            }
          } else {
            if (co_ok) {
              if constexpr (CAN_CORESIDE) k_accumulate<C, 2, true><<<gridn, 128, 0, sq>>>(ACC_ARGS);
            } else if (minw == 2) k_accumulate<C, 2><<<gridn, 128, 0, sq>>>(ACC_ARGS);
            else if (minw == 4) k_accumulate<C, 4><<<gridn, 128, 0, sq>>>(ACC_ARGS);
            else if (minw == 1) k_accumulate<C, 1><<<gridn, 128, 0, sq>>>(ACC_ARGS);
            else k_accumulate<C, 3><<<gridn, 128, 0, sq>>>(ACC_ARGS);
          }
#undef ACC_ARGS
        }
        LAUNCH_CHECK("k_accumulate", sq);
        KernelTimer::end(0, sq);
        k_fold_overflow<C><<<std::min<uint32_t>(ocap, 4096), 64, 0, sq>>>(buckets, ovfpart, ovf, d_firsts.as<uint32_t>() + g_ovf_off[g] + g, ovfcnt, ocap);
        LAUNCH_CHECK("k_fold_overflow", sq);
        return ICICLE_SUCCESS;
      };

      // ---- phase 3 of a group: bucket reduction of its windows, then their share of the window combine
      auto reduce_group = [&](int g, hipStream_t sq, uint32_t seg_lo, uint32_t nsegr) -> icicle_error_t {
        const int w0 = glo[g], nw = ghi[g] - glo[g];
        // windows [w0, split) of the group are the narrow windows of a mixed-width plan (half the buckets, nseg_lo chunks), the
        // rest wide; one launch of each kernel serves both (a uniform plan has n_lo = 0: no narrow windows). (bb == 1 whenever n_lo > 0)
        const int split = std::min(std::max(pl.n_lo, w0), w0 + nw);
        const uint32_t nlo_w = (uint32_t)(split - w0);
        // narrow windows: half the rows per lane, so that they have as many chunks as the wide ones (see k_reduce_wave)
        // Measured and left OFF (profiles/r06_msm_reduce_balance_ab.txt, same box, BN254 2^26): tail 4.72 / 4.75 ms without, 4.90 / 4.86 ms
        // with -- the extra 1280 waves bring 19 scan additions each, more than the evened-out SIMD load gives back. ICICLE_HIP_MSM_REDUCE_BALANCE=1: on.
        static const bool balance = getenv("ICICLE_HIP_MSM_REDUCE_BALANCE") && atoi(getenv("ICICLE_HIP_MSM_REDUCE_BALANCE")) != 0;
        const uint32_t mrow_lo = (balance && nlo_w > 0 && mrow >= 2 && !hook) ? mrow / 2 : mrow;
        const uint32_t nseg_lo = std::max<uint32_t>(1, (nb / 2) / (64 * mrow_lo));
        const uint32_t log_chunk_lo = mrow_lo == mrow ? log_chunk : log_chunk - 1;
        const bool direct = (nseg == 1 && nsegr == 1 && seg_lo == 0);
        typename E::Proj* win = d_win.as<typename E::Proj>() + w0;
        typename E::Proj* cV = chunkV + (size_t)w0 * nseg;
        typename E::Proj* cT = chunkT + (size_t)w0 * nseg;
        const size_t nblocks = (size_t)nlo_w * nseg_lo + (size_t)(nw - nlo_w) * nsegr;
        // whole small windows: several per wave (k_reduce_small); rows per lane: 8, or 4 for the tiniest windows
        static const bool small_on = !(getenv("ICICLE_HIP_MSM_REDUCE_SMALL") && atoi(getenv("ICICLE_HIP_MSM_REDUCE_SMALL")) == 0);
        // Four lanes per bucket column while the reduction is a latency chain (single MSM, uniform plan, one device): the window kernel
        // whenever a window has at most 64 chunks, the wave kernel when its shorter additions beat the one-lane kernel's 3.5 x fewer
        // lanes -- up to ~2^18 terms; beyond, the reduction is work and the one-lane kernel wins (2^20: 0.56 against 0.43 ms).
        // ICICLE_HIP_MSM_REDUCE_QUAD=0: off (A/B).
        static const bool quad_on = !(getenv("ICICLE_HIP_MSM_REDUCE_QUAD") && atoi(getenv("ICICLE_HIP_MSM_REDUCE_QUAD")) == 0);
        bool quad_done = false, quad_window = false;
        // (G1, and the G2 whose point fits the registers four lanes deep: BN254's -- BLS12-381's 28-limb Fq2 elements do not)
        constexpr bool QUAD_OK = C::EXT_DEGREE == 1 || sizeof(typename E::XYZZ) <= 288;
        if constexpr (QUAD_OK) {
          const bool small_path = small_on && direct && nlo_w == 0 && nb >= 16 && nb <= 512 && nw >= 64;
          if (quad_on && !hook && NG == 1 && bb == 1 && pl.n_lo == 0 && seg_lo == 0 && nsegr == nseg && nb >= 16 && !small_path) {
            quad_window = nseg <= 64 && !direct;
            if (nb <= 65536) {
              // chunks per window so that the waves fill the SIMDs once (a wave past 1024 doubles the time of its SIMD), rows to match
              uint32_t nsq = std::min<uint32_t>(std::min<uint32_t>(64, std::max<uint32_t>(1, 1024 / (uint32_t)nw)), nb / 16);
              const uint32_t mq = (nb + 16 * nsq - 1) / (16 * nsq);
              nsq = (nb + 16 * mq - 1) / (16 * mq);
              // additions on the chain x instructions per addition (in hundreds) x rounds of waves on the 1024 SIMDs
              const uint64_t cost_q = (uint64_t)(2 * mq + 13) * 13 * (((uint64_t)nw * nsq + 1023) / 1024);
              const uint64_t cost_s = (uint64_t)(2 * mrow + 19) * 29 * (((uint64_t)nw * nseg + 1023) / 1024);
              if (cost_q < cost_s) {
                const bool direct_q = nsq == 1;
                k_reduce_wave_quad<C><<<(unsigned)((size_t)nw * nsq), 64, 0, sq>>>(buckets + (size_t)w0 * nb, chunkV, chunkT, direct_q ? win : nullptr, nb, mq, nsq);
                LAUNCH_CHECK("k_reduce_wave_quad", sq);
                if (!direct_q) {
                  unsigned qthreads = 64;
                  while (qthreads < 4 * nsq)
                    qthreads <<= 1;
                  k_reduce_window_quad<C><<<(unsigned)nw, qthreads, 0, sq>>>(chunkV, chunkT, win, nsq, 16 * mq);
                  LAUNCH_CHECK("k_reduce_window_quad", sq);
                }
                quad_done = true;
              }
            }
          }
        }
        if (quad_done) {
        } else if (small_on && direct && nlo_w == 0 && nb >= 16 && nb <= 512 && nw >= 64) {
          uint32_t lw_log = 2;
          while ((nb >> lw_log) > 8 && lw_log < 5)
            lw_log++;
          const uint32_t per_wave = 64u >> lw_log;
          k_reduce_small<C><<<(unsigned)(((size_t)nw + per_wave - 1) / per_wave), 64, 0, sq>>>(buckets + (size_t)w0 * nb, win, nb, lw_log, (uint32_t)nw);
          LAUNCH_CHECK("k_reduce_small", sq);
        } else if (nblocks) {
          k_reduce_wave<C><<<(unsigned)nblocks, 64, 0, sq>>>(buckets + (size_t)w0 * nb, cV, cT, direct ? win : nullptr, nb, nb, mrow, seg_lo, nsegr, nlo_w, nseg_lo, mrow_lo);
          LAUNCH_CHECK("k_reduce_wave", sq);
        }
        if (!direct && !quad_done && quad_window) {
          if constexpr (QUAD_OK) {
            unsigned qthreads = 64;
            while (qthreads < 4 * nseg)
              qthreads <<= 1;
            k_reduce_window_quad<C><<<(unsigned)nw, qthreads, 0, sq>>>(cV, cT, win, nseg, 64 * mrow);
            LAUNCH_CHECK("k_reduce_window_quad", sq);
          }
        } else if (!direct && !quad_done) {
          unsigned rthreads = 64; // power of two (LDS tree), >= the chunks of a window
          while (rthreads < std::max(nsegr, nlo_w ? nseg_lo : 0u) && rthreads < RWL)
            rthreads <<= 1;
          k_reduce_window<C><<<(unsigned)nw, rthreads, 0, sq>>>(cV, cT, win, nsegr, seg_lo, log_chunk, nlo_w, nseg_lo, log_chunk_lo);
          LAUNCH_CHECK("k_reduce_window", sq);
        }
        if (NG > 1) { // (bb == 1) this group's windows, scaled, into its partial slots
          const int nbw = (nw + FINAL_WINDOWS_PER_BLOCK - 1) / FINAL_WINDOWS_PER_BLOCK;
          k_final<C><<<dim3((unsigned)nbw, 1), 64, 0, sq>>>(d_win.as<typename E::Proj>(), nullptr, wpf, ww, w0, nw, d_part.as<typename E::Proj>(), part_slot[g], 1);
          LAUNCH_CHECK("k_final(group)", sq);
        }
        return ICICLE_SUCCESS;
      };

      if (NG == 1) {
        ICICLE_TRY(sort_group(0, st));
        KernelTimer::end(2, st);
        ICICLE_TRY(accumulate_group(0, st));
        KernelTimer::begin(3, st);
        uint32_t seg_lo = 0, nsegr = nseg;
        if (hook) {
          bool skip = false;
          ICICLE_TRY(hook->after_accumulate(buckets, tw, nb, nseg, m, st, &skip, &seg_lo, &nsegr));
          if (skip) continue; // another shard of this device (or the exchange step) produces the result
        }
        ICICLE_TRY(reduce_group(0, st, seg_lo, nsegr));
        // Window combine on the HOST when the result goes there anyway (every wrapper's default; the reference's phase 3 is
        // host code too, cpu_msm.hpp:365-417): result = sum_w 2^offset(w) S_w is a chain of ~250 DEPENDENT doublings whatever
        // the window size. On the GPU that is the latency floor of a small MSM (k_final: 0.65 ms with four lanes per window);
        // one host core runs the same Jacobian chain (ec.hpp dbl_jac, 2M + 5S per step) in ~0.15 ms, and the 13 window sums
        // are a 1.4 KB download that replaces the download of the result. Single MSMs only (a batch of 1024 combines runs in
        // parallel on the GPU); device-resident results keep k_final. ICICLE_HIP_MSM_HOST_COMBINE=0: always k_final.
        static const bool host_combine_env = !(getenv("ICICLE_HIP_MSM_HOST_COMBINE") && atoi(getenv("ICICLE_HIP_MSM_HOST_COMBINE")) == 0);
        // (a device-resident result of a SYNCHRONOUS call takes the same route -- the call waits for the stream anyway -- and the 96 / 144
        //  bytes go back up: 2^20 2.99 -> see profiles/r06_notes.md; an asynchronous call must not block, so it keeps k_final)
        if (host_combine_env && (!cfg->are_results_on_device || !cfg->is_async) && batch == 1 && bb == 1 && !hook && wpf <= 64) {
          typename E::Proj hw[64];
          HIP_TRY(hipMemcpyAsync(hw, d_win.ptr(), (size_t)wpf * sizeof(typename E::Proj), hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
          KernelTimer::end(3, st);
          HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
          typename E::Proj acc = hw[wpf - 1];
          for (int w = wpf - 2; w >= 0; w--) { // Horner from the top window: acc = 2^width(w) * acc + S_w
            typename E::Jac j = E::to_jac(acc);
            for (int i = 0; i < pl.width(w); i++)
              j = E::dbl_jac(j);
            acc = E::add(E::from_jac(j), hw[w]);
          }
          if (cfg->are_results_on_device) {
            uint32_t words[RW];
            E::store_proj_canonical(words, acc);
            HIP_TRY(hipMemcpyAsync(results_v, words, sizeof(words), hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
            HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
          } else {
            E::store_proj_canonical((uint32_t*)results_v, acc);
          }
          return ICICLE_SUCCESS;
        }
        static const bool horner_on = !(getenv("ICICLE_HIP_MSM_BATCH_HORNER") && atoi(getenv("ICICLE_HIP_MSM_BATCH_HORNER")) == 0);
        if (horner_on && bb >= 64 && wpf >= 2) {
          k_final_horner<C><<<(unsigned)((bb + 15) / 16), 64, 0, st>>>(d_win.as<typename E::Proj>(), d_res + (size_t)b0 * RW, wpf, ww, bb);
          LAUNCH_CHECK("k_final_horner", st);
        } else {
          const int nbw = (wpf + FINAL_WINDOWS_PER_BLOCK - 1) / FINAL_WINDOWS_PER_BLOCK;
          k_final<C><<<dim3((unsigned)nbw, (unsigned)bb), 64, 0, st>>>(d_win.as<typename E::Proj>(), d_res + (size_t)b0 * RW, wpf, ww, 0, wpf, nbw > 1 ? d_part.as<typename E::Proj>() : nullptr, 0, bb);
          LAUNCH_CHECK("k_final", st);
          if (nbw > 1) { // more than 16 windows: one single-wave block per 16 of them, then the sum of the partials
            k_final_combine<C><<<(unsigned)((bb + 63) / 64), 64, 0, st>>>(d_part.as<typename E::Proj>(), d_res + (size_t)b0 * RW, nbw, bb);
            LAUNCH_CHECK("k_final_combine", st);
          }
        }
        KernelTimer::end(3, st);
      } else {
        // main stream: sort(0), accumulate(0), accumulate(1), ... ; sort stream: sort(1), sort(2), ... ; reduce stream:
        // reduce(0), reduce(1), ... each behind its accumulation. The main stream finally waits for the reduce stream.
        for (int g = 0; g < NG; g++) {
          ev_sorted[g] = ring_event();
          ev_acc[g] = ring_event();
          if (!ev_sorted[g] || !ev_acc[g]) return ICICLE_ALLOCATION_FAILED;
        }
        ICICLE_TRY(sort_group(0, st));
        KernelTimer::end(2, st); // (the exposed part of the sort)
        if (co_ok) {
          // the co-resident sort of group 1 starts WITH the accumulation of group 0, not beside the stand-alone sort of
          // group 0 (where it would only take bandwidth from it and delay the first accumulation)
          hipEvent_t ev_s0 = ring_event();
          if (!ev_s0) return ICICLE_ALLOCATION_FAILED;
          HIP_TRY(hipEventRecord(ev_s0, st), ICICLE_SYNCHRONIZATION_FAILED);
          HIP_TRY(hipStreamWaitEvent(s_sort, ev_s0, 0), ICICLE_SYNCHRONIZATION_FAILED);
        }
        for (int g = 1; g < NG; g++) {
          ICICLE_TRY(sort_group(g, s_sort));
          HIP_TRY(hipEventRecord(ev_sorted[g], s_sort), ICICLE_SYNCHRONIZATION_FAILED);
        }
        for (int g = 0; g < NG; g++) {
          if (g > 0) HIP_TRY(hipStreamWaitEvent(s_acc, ev_sorted[g], 0), ICICLE_SYNCHRONIZATION_FAILED);
          ICICLE_TRY(accumulate_group(g, s_acc));
          HIP_TRY(hipEventRecord(ev_acc[g], s_acc), ICICLE_SYNCHRONIZATION_FAILED);
          if (g < NG - 1) {
            HIP_TRY(hipStreamWaitEvent(s_red, ev_acc[g], 0), ICICLE_SYNCHRONIZATION_FAILED);
            ICICLE_TRY(reduce_group(g, s_red, 0, nseg));
          }
        }
        KernelTimer::begin(3, st); // (the exposed tail: the last group's reduction + the combine)
        ICICLE_TRY(reduce_group(NG - 1, st, 0, nseg));
        hipEvent_t ev_tail = ring_event();
        if (!ev_tail) return ICICLE_ALLOCATION_FAILED;
        HIP_TRY(hipEventRecord(ev_tail, s_red), ICICLE_SYNCHRONIZATION_FAILED);
        HIP_TRY(hipStreamWaitEvent(st, ev_tail, 0), ICICLE_SYNCHRONIZATION_FAILED);
        k_final_combine<C><<<1, 64, 0, st>>>(d_part.as<typename E::Proj>(), d_res + (size_t)b0 * RW, part_slots, bb);
        LAUNCH_CHECK("k_final_combine", st);
        KernelTimer::end(3, st);
      }
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

} // namespace icicle_hip
#include "msm_multi.hpp"
namespace icicle_hip {

  // <curve>_msm_precompute_bases(input_bases, nof_bases, config, output_bases). `nof_bases` is the number of bases of ONE
  // MSM: when every MSM of a batch brings its own (batch_size > 1, are_points_shared_in_batch = false) the input holds
  // nof_bases * batch_size bases and all of them are extended -- this is what both wrappers hand over
  // (wrappers/rust/icicle-core/src/msm/mod.rs:296 `points.len() / config.batch_size`, wrappers/golang/curves/bn254/msm/
  // msm.go:39-43) and what their tests rely on (msm/tests.rs:195-217). The reference's CPU backend extends nof_bases
  // points whatever the batch (cpu_msm.hpp:470-485) and leaves the rest of the table unwritten; rounds 1-4 here took
  // nof_bases for the total. Table layout as the reference's: output[pf * i + j] = 2^(j * c * wpf) * input[i].
  template <class C>
  static icicle_error_t msm_precompute_run(const void* in_v, int n_one, const icicle_msm_config_t* cfg, void* out_v)
  {
    using E = EC<C>;
    constexpr int PW = 2 * E::N32;
    if (!cfg || n_one < 0) return ICICLE_INVALID_ARGUMENT;
    if (n_one == 0) return ICICLE_SUCCESS;
    if (!in_v || !out_v) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const int batch = std::max(1, cfg->batch_size);
    const long long n_all = (!cfg->are_points_shared_in_batch && batch > 1) ? (long long)n_one * batch : n_one;
    const MsmPlan pl = make_plan(n_one, C::fr::NBITS, *cfg);
    const int pf = pl.pf;
    if (n_all * pf >= (1ll << 31)) return ICICLE_INVALID_ARGUMENT;
    const int n = (int)n_all;
    TempBuf d_in_tmp, d_out_tmp;
    const uint32_t* d_in = (const uint32_t*)in_v;
    uint32_t* d_out = (uint32_t*)out_v;
    // input location: are_points_on_device ; output location: are_results_on_device (msm.h:39-47) -- as the Go wrapper sets
    // them (wrappers/golang/core/msm.go:138-139). The Rust wrapper hands its MSMConfig through untouched
    // (icicle-core/src/msm/mod.rs:288-302: the two flags are private and only msm() derives them) while the output is a
    // DeviceSlice by type and the input may be one: a flag left at "host" is checked against the pointer itself.
    const bool in_on_device = cfg->are_points_on_device || points_to_device_memory(in_v);
    const bool out_on_device = cfg->are_results_on_device || points_to_device_memory(out_v);
    // nof_bases is the size of ONE MSM (see above): with per-MSM bases this call reads and writes batch_size times that. A caller
    // that meant nof_bases as the TOTAL (the reading the reference's CPU backend takes, cpu_msm.hpp:470-485, and rounds 1-4 of this
    // backend took) would be read and written batch_size times past its buffers: refused where the device allocation proves it
    // (ADVICE r05). INTEGRATION.md states the convention.
    if ((in_on_device && overruns_device_allocation(in_v, (size_t)n * PW * 4)) || (out_on_device && overruns_device_allocation(out_v, (size_t)n * pf * PW * 4))) {
      fprintf(stderr, "[icicle_hip] msm_precompute_bases: nof_bases = %d x batch_size %d (bases per MSM, are_points_shared_in_batch = false) x precompute_factor %d does not fit the device buffers handed in\n", n_one, batch, pf);
      return ICICLE_INVALID_ARGUMENT;
    }
    if (!in_on_device) {
      HIP_TRY(d_in_tmp.alloc((size_t)n * PW * 4, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(d_in_tmp.ptr(), in_v, (size_t)n * PW * 4, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      d_in = d_in_tmp.as<uint32_t>();
    }
    if (!out_on_device) {
      HIP_TRY(d_out_tmp.alloc((size_t)n * pf * PW * 4, st), ICICLE_ALLOCATION_FAILED);
      d_out = d_out_tmp.as<uint32_t>();
    }
    // register budget of the two kernels (waves per SIMD the allocator must leave room for): a doubling chain keeps three
    // coordinates and four or five temporaries: 3 waves for 9-limb fields, 2 for 14-limb fields and BN254's Fq2, 1 for BLS12's Fq2
    constexpr bool BIGPT = sizeof(typename E::XYZZ) > 256; // G2
    constexpr int MINW = BIGPT ? (sizeof(typename E::XYZZ) <= 288 ? 2 : 1) : (E::F::N <= 9 ? 3 : 2);
    const bool in16 = ((uintptr_t)d_in & 15) == 0, out16 = ((uintptr_t)d_out & 15) == 0;
    if (pf == 1) {
      HIP_TRY(hipMemcpyAsync(d_out, d_in, (size_t)n * PW * 4, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
    } else {
      // side buffers (Z and the prefix products) for one chunk of bases: at most ~1 GiB together, at least a chip-filling launch
      const size_t per_base = (size_t)(pf - 1) * E::N32 * 4 * 2;
      long long chunk = std::max<long long>(256 * 1024, (long long)((1ull << 30) / per_base));
      chunk = std::min<long long>(chunk, n);
      TempBuf side;
      HIP_TRY(side.alloc((size_t)chunk * per_base, st), ICICLE_ALLOCATION_FAILED);
      uint32_t* zbuf = side.as<uint32_t>();
      uint32_t* pbuf = zbuf + (size_t)chunk * (pf - 1) * E::N32;
      const int shift = pl.c * pl.wpf;
      for (long long i0 = 0; i0 < n; i0 += chunk) {
        const int cnt = (int)std::min<long long>(chunk, n - i0);
        k_precompute_chains<C, MINW><<<(unsigned)((cnt + 63) / 64), 64, 0, st>>>(d_in, d_out, zbuf, i0, cnt, pf, shift, cfg->are_points_montgomery_form, in16, out16);
        LAUNCH_CHECK("k_precompute_chains", st);
        const long long nq = (long long)cnt * (pf - 1);
        const int run = nq >= (long long)PRECOMP_RUN_LARGE * 256 * 1024 ? PRECOMP_RUN_LARGE : PRECOMP_RUN_SMALL;
        const long long runs = (nq + run - 1) / run;
        k_precompute_affine<C, MINW><<<(unsigned)((runs + 63) / 64), 64, 0, st>>>(d_out, zbuf, pbuf, i0, nq, pf, cfg->are_points_montgomery_form, out16, run);
        LAUNCH_CHECK("k_precompute_affine", st);
      }
    }
    HIP_TRY(hipGetLastError(), ICICLE_INVALID_ARGUMENT);
    if (!out_on_device) {
      HIP_TRY(hipMemcpyAsync(out_v, d_out, (size_t)n * pf * PW * 4, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    } else if (!cfg->is_async) {
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    }
    table_register(out_v, (size_t)n * pf * PW * 4, (size_t)PW * 4, pf, pl.c); // msm() on this table finds its window size here
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

