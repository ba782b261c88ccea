// Register-blocked pass kernel of the 31-bit NTT (BabyBear, KoalaBear): shared by ntt.hip (row-major batches of
// base-field transforms, LN = false) and ntt_lanes.hip (interleaved transforms -- columns_batch and the quartic
// extension field --, LN = true), so that the two sets of instantiations compile in parallel.
#pragma once
#include "common.h"
#include "smallfield.hpp"
#include "ntt_plan.h"
#include <algorithm>

namespace icicle_hip {

  // ---- fast pass: register-blocked radix-2 butterflies, 4 stages per LDS round trip -------------
  // Every ordering, coset or not, any batch layout, both directions. Bit-reversed INPUT (kRN) is consumed natively by the RN
  // variants below (round 5); kRR and forward cosets on reversed input still arrive behind the reordering pre-pass, and only
  // single-pass transforms with reversed input are left to the generic kernel.
  // A pass computes 2^s-point transforms on a [L x T] tile (T adjacent columns => every HBM access is
  // a run of T contiguous words). A thread owns 16 tile elements per "round" and runs up to 4 butterfly
  // stages on them in registers; the FIRST round loads its operands straight from HBM and the LAST
  // round stores straight to HBM, so s = 8 costs one LDS round trip and one barrier per row-transform.
  // Stage twiddles and the inter-pass twiddle w_M^(j*K) do not depend on the batch row: they are
  // loaded ONCE per block into registers and the block loops over `rows_per_block` rows
  // (double-buffered LDS).
  //   DIF == false: column pass (HBM-contiguous direction = tile column t on both sides). The first
  //                 round gathers bit-reversed rows, stages run DIT (ascending), output is natural.
  //   DIF == true : row pass = the last pass (HBM-contiguous direction = k on the way in, t on the
  //                 way out). Natural order in, stages run DIF (descending); the first executed round
  //                 maps lanes along k, later rounds along t, so both HBM sides stay coalesced and the
  //                 digit-reversal to natural order happens in the store addresses. INV folds 1/N in.
  // Template: NQ0 = stages in the lowest round (1..4), NR = rounds; s = NQ0 + 4*(NR-1).
  template <class S, int NQ, bool DIF, bool SKIP_TRIVIAL>
  __device__ __forceinline__ void ntt_stages(uint32_t* x, const uint32_t* w)
  {
#pragma unroll
    for (int jj = 0; jj < NQ; jj++) {
      const int j = DIF ? (NQ - 1 - jj) : jj;
      const int half = 1 << j;
#pragma unroll
      for (int bfly = 0; bfly < (1 << NQ) / 2; bfly++) {
        const int pos = bfly & (half - 1);
        const int i = ((bfly >> j) << (j + 1)) + pos;
        const bool trivial = SKIP_TRIVIAL && pos == 0; // w_L^0 = 1
        if (DIF) {
          const uint32_t sum = S::add(x[i], x[i + half]);
          const uint32_t dif = S::sub(x[i], x[i + half]);
          x[i] = sum;
          x[i + half] = trivial ? dif : S::mul(dif, w[half - 1 + pos]);
        } else {
          const uint32_t v = trivial ? x[i + half] : S::mul(x[i + half], w[half - 1 + pos]);
          const uint32_t uu = x[i];
          x[i] = S::add(uu, v);
          x[i + half] = S::sub(uu, v);
        }
      }
    }
  }

  template <int BITS>
  __device__ __forceinline__ constexpr uint32_t brev_c(uint32_t v)
  {
    uint32_t r = 0;
    for (int i = 0; i < BITS; i++)
      r |= ((v >> i) & 1u) << (BITS - 1 - i);
    return r;
  }

  // waves per SIMD the register allocator must leave room for: the two-round variants are LDS-limited to 4 waves
  // per SIMD (2 blocks of 8 waves per CU) and fit 128 VGPRs; the three-round, coset and bit-reversed-output
  // variants (separate instantiations, so that they do not cost the plain path registers) get 2 (the kNR store needs
  // ~170-185 VGPRs: held to 128 it spilled 236-332 bytes per thread through rounds 1-3, tools/kernel_regs.py)
  // Round 6: the plain three-round ROW pass (the 512-row last pass of a 2^27-point transform) took 148 VGPRs under a budget of two waves
  // per SIMD = ONE 512-thread block per CU; held to four waves it compiles to 112 VGPRs without scratch and two blocks fit (the
  // three-round column pass always did: 120). The coset / bit-reversed-input / bit-reversed-output three-round variants spill at 128
  // (92 - 132 B, tools/kernel_regs.py) and keep two.
  constexpr int ntt_fast_min_waves(int nr, bool extra, bool plain_row = false) { return ((nr <= 2 || plain_row) && !extra) ? 4 : 2; }

  // 4x4 transpose across the four 16-lane rows of a wave: lanes {l, l+16, l+32, l+48} form a group, in: the lane of
  // row r holds v[c] = M[r][c]; out: v[c] = M[c][r]. gfx950's v_permlane16_swap (odd rows of the first operand <->
  // even rows of the second) and v_permlane32_swap (upper half of the first <-> lower half of the second) are exactly
  // the two exchange steps of the transpose: four instructions, no selects.
  typedef uint32_t ntt_u2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ void rows_transpose(uint32_t* v)
  {
    const ntt_u2 a01 = __builtin_amdgcn_permlane16_swap(v[0], v[1], false, false);
    const ntt_u2 a23 = __builtin_amdgcn_permlane16_swap(v[2], v[3], false, false);
    const ntt_u2 w02 = __builtin_amdgcn_permlane32_swap(a01.x, a23.x, false, false);
    const ntt_u2 w13 = __builtin_amdgcn_permlane32_swap(a01.y, a23.y, false, false);
    v[0] = w02.x, v[1] = w13.x, v[2] = w02.y, v[3] = w13.y;
  }

  // V4 (16-byte lanes, row pass): with unit element stride, 32-column tiles and radix-16 rounds a thread's 16
  // operands / results sit in 16 different rows of ONE column. Lane l of a wave works for output column
  // 4*(l & 7) + (l >> 4) of group 2*wave + ((l >> 3) & 1) (and for input row 4*(l & 3) + (l >> 4) of its 16-row block):
  // the four lanes {l, l+16, l+32, l+48} own four ADJACENT columns of the same rows. The lane in wave-row r moves the
  // rows {4g + r} as uint4, eight consecutive lanes cover one 128-byte line, and rows_transpose() hands every word to
  // its owner: a quarter of the memory instructions for 4 extra VALU instructions per 4 words. Measured (same box,
  // profiles/r02_notes.md section 6): the row pass of 2^24 x 64 1.90 -> 1.83 ms, 2^16 x 1024 transforms -7 %. The same
  // paths on the column passes were built and measured neutral (+-1 %), so they are not in the tree. The host picks
  // V4 when every base is 16-byte aligned; results are identical word for word.
  // BIG: 1024-thread blocks for the 512- and 1024-row column passes of transforms of 2^27 points and more, so that their
  // tiles are 32 (resp. 16) columns wide instead of 16 (8): 128-byte HBM runs instead of 64-byte ones (a 64-byte-run tile
  // copy moves 3.9 TB/s, a 128-byte-run one 5.0: profiles/r03_notes.md section 2). The three-round column pass needs 120
  // VGPRs, which is what a 16-wave block leaves per lane.
  // LN (lane-native tiles): the buffer interleaves `ltot` transforms word by word -- element j of transform l at word
  // j * es + l (columns_batch: es = ltot = batch * lanes, ntt_cpu.h:250,274-275; the quartic extension field in a row-major
  // batch: es = ltot = 4). The tile's 32 word-columns are then (T logical columns) x (TL = 2^lsh interleaved
  // transforms), lanes fastest, so every HBM access of every pass is still a run of 32 contiguous words: with TL = 32 a
  // tile is ONE logical column of 32 adjacent transforms (no transposition left for the last pass to do), with TL = 4
  // eight columns of one extension-field row. A "row" of the launch is one slice of TL transforms; the lanes of a
  // partial last slice (ltot not a multiple of TL) load a clamped address and store nothing.
  // RN (bit-reversed INPUT consumed natively, kRN; round 5): in-place DIT by digits. Every pass reads the rows of its digit as
  // they lie in memory (bit-reversed) and writes them back in natural order, so after the last pass the whole transform is in
  // natural order -- P reads + P writes like kNN, no reordering pre-pass (the reference reads through a permutation too,
  // ntt_cpu.h:252,286-296). Pass 0 works on the contiguous runs of 2^s0 elements that hold digit j_0: RN == 2, lanes along
  // the run on both sides (16 consecutive words per thread in, 4-byte coalesced out), column-major LDS tile padded by one
  // word per 16; passes q >= 1 (and pass 0 of lane-native tiles, whose word-columns are interleaved transforms) are column
  // passes, RN == 1: the DIT column code with direct loads and the factor w_M^(j_{q+1} (column + A_q row)) behind it.
  // The last pass applies 1/N (or g^-k / N) instead. Forward cosets and kRR keep the pre-pass.
  template <class PR, int NQ0, int NR, bool DIF, bool INV, bool COSET, bool OUTREV, bool V4 = false, bool BIG = false, bool LN = false, int RN = 0>
  __global__ __launch_bounds__(BIG ? 1024 : 512, BIG ? 1 : ntt_fast_min_waves(NR, (COSET && (DIF || LN || RN != 0)) || OUTREV, DIF && !COSET && RN == 0 && !LN)) void k_ntt_fast(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ tw, const uint32_t* __restrict__ ctab, PassDesc pd, NttLaunch nl, uint32_t rows_per_block)
  {
    using S = SmallField<PR>;
    constexpr int SS = NQ0 + 4 * (NR - 1);
    constexpr uint32_t L = 1u << SS;
    constexpr int E = (SS >= 4) ? 16 : (1 << SS); // elements per thread per round
    constexpr int G0 = E >> NQ0;                  // groups per thread in the lowest round
    constexpr uint32_t NG16 = L / E;              // threads per tile column
    constexpr int QT = NR >= 2 ? NQ0 + 4 * (NR - 2) : 0; // first stage of the top round
    constexpr int KB_BITS = SS - NQ0;
    // (Round 6, measured and removed: generating the 16 inter-pass factors of the plain three-round COLUMN pass at the store -- one running
    //  product instead of 16 registers -- to pay for the cross-row prefetch the two-round passes have. At 128 VGPRs the kernel spilled 84 B
    //  and 2^26 x 16 went from 6.72 to 7.91 ms, 2^25 x 32 from 6.09 to 6.50: profiles/r06_notes.md section 4b.)
    extern __shared__ uint32_t lds[];
    static_assert(!LN || (!V4 && !BIG), "lane-native tiles: 4-byte lanes, 512-thread blocks");
    static_assert(RN == 0 || (!DIF && !INV && !OUTREV && !V4 && !BIG), "bit-reversed input: DIT column-type passes (the direction is a run-time flag there)");
    static_assert(RN != 2 || !LN, "the run pass is the row-major form of pass 0");
    const uint32_t lsh = LN ? nl.lsh : 0u, lmask = (1u << lsh) - 1u;
    const uint32_t TC = pd.T;      // logical columns per tile (LN with column groups: per group of nl.cgrp tile-rows)
    const uint32_t cgrp = LN ? nl.cgrp : 1u, agrp = LN ? nl.agrp : 1u;
    const uint32_t T = (LN ? nl.tcl : TC) << lsh; // word-columns per tile = LDS row length = threads along t
    const uint32_t TP = T + 1;
    // Narrow tiles (T*4 B < one 128 B line): neighbouring tiles share HBM lines. Workgroups are dealt
    // round-robin to the 8 XCDs, so give each XCD a contiguous range of tiles -- then the tiles that
    // share a line run back to back on the same XCD and meet in its L2.
    uint32_t tile = blockIdx.x;
    if (pd.xcd_remap) tile = (blockIdx.x & 7u) * (pd.ntiles >> 3) + (blockIdx.x >> 3);
    const uint32_t a = (tile / pd.tiles_per_a) * agrp, ct = tile % pd.tiles_per_a; // (agrp > 1: the first of the block's outer indices)
    const uint64_t in_base = (uint64_t)a * pd.in_base_a + (uint64_t)ct * pd.in_base_ct;
    const uint64_t max_mask = ((uint64_t)1 << nl.log_max) - 1;
    const uint32_t lstride_log = nl.log_max - SS;
    // mapping B (lanes along t) everywhere except the first executed round of a DIF pass (mapping A)
    static_assert(!V4 || (DIF && NQ0 == 4 && NR >= 2 && !OUTREV), "16-byte lanes: row pass with radix-16 rounds only");
    // V4 (T == 32): lane l -> column 4*(l & 7) + (l >> 4), group 2*wave + ((l >> 3) & 1); mapping A (NG16 % 16 == 0):
    // row 4*(l & 3) + (l >> 4) of its 16-row block, column 4*wave + ((l >> 2) & 3) -- see the V4 note above
    const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t wrow = ln >> 4, wrev = ((wrow & 1u) << 1) | (wrow >> 1); // wave-row of this lane, 2-bit reversed
    const uint32_t tB = V4 ? 4u * (ln & 7u) + wrow : threadIdx.x % T;
    const uint32_t gB = V4 ? 2u * wv + ((ln >> 3) & 1u) : threadIdx.x / T;
    // LN: lanes fastest in both mappings; mapping A = (lane, k-group, logical column)
    const uint32_t gA = V4 ? (threadIdx.x % NG16 & ~15u) + 4u * (ln & 3u) + wrow : (threadIdx.x >> lsh) % NG16;
    const uint32_t cA = V4 ? (threadIdx.x / NG16 & ~3u) + ((ln >> 2) & 3u) : threadIdx.x / (NG16 << lsh); // logical column, mapping A
    const uint32_t tA = LN ? ((cA << lsh) | (threadIdx.x & lmask)) : cA;                                   // LDS column, mapping A
    const uint32_t cB = tB >> lsh;                                                                         // logical column, mapping B
    const uint32_t lB = tB & lmask, lA = threadIdx.x & lmask;                                              // lane inside the slice

    auto tw_load = [&](uint64_t idx) -> uint32_t {
      idx &= max_mask;
      if (nl.inverse) idx = (((uint64_t)1 << nl.log_max) - idx) & max_mask;
      return tw[idx];
    };

    // coset factor g^e (forward) or N^-1 * g^-e (inverse) from the two-level table: lo[e & 4095] * hi[e >> 12]
    auto cpow = [&](uint64_t e) -> uint32_t { return S::mul(ctab[e & 4095], ctab[4096 + (e >> 12)]); };
    // COSET variants are separate instantiations: the 16 per-thread factors must not cost the plain path registers
    const bool coset_in = RN == 0 && COSET && !nl.inverse && pd.pidx == 0;  // x[j] *= g^j on the way in (first pass)
    const bool coset_out = COSET && nl.inverse && pd.is_last;    // X[k] *= g^-k / N on the way out (last pass)

    // ---- per-thread twiddles, loaded once per block ---------------------------------------------
    uint32_t w0[(1 << NQ0)]; // lowest round: group-independent (wave-uniform => scalar registers)
#pragma unroll
    for (int j = 0; j < NQ0; j++)
#pragma unroll
      for (int pos = 0; pos < (1 << j); pos++)
        w0[(1 << j) - 1 + pos] = tw_load(((uint64_t)pos << (SS - 1 - j)) << lstride_log);
    uint32_t wr[NR > 1 ? NR - 1 : 1][16];
#pragma unroll
    for (int r = 1; r < NR; r++) {
      const int q0 = NQ0 + 4 * (r - 1);
      const uint32_t g = (RN == 2 || (DIF && r == NR - 1)) ? gA : gB;
      const uint32_t base_low = g & ((1u << q0) - 1);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int pos = 0; pos < (1 << j); pos++)
          wr[r - 1][(1 << j) - 1 + pos] = tw_load(((uint64_t)(base_low + ((uint32_t)pos << q0)) << (SS - 1 - (q0 + j))) << lstride_log);
    }
    // base row of this thread's elements in the top round (natural order, mapping B)
    const uint32_t baseT = (NR == 1) ? 0u : (((gB >> QT) << (QT + 4)) | (gB & ((1u << QT) - 1)));
    uint32_t wip[E]; // inter-pass twiddles of the E elements this thread stores (column passes)
    bool rn_fac = false; // RN: whether the stored values are multiplied by wip[] at all (not in a forward last pass)
    // RN with column groups (lane-native, few slices): the launch rows of a block are cgrp ADJACENT logical columns (runs in pass
    // 0). The stage twiddles are shared; the factor behind the pass is not -- w^(j_next (column + A row)) with the row's own
    // column, and in pass 0 the row's own j_next -- so each row fetches the first term and the ratio of its geometric
    // progression with its operands (load_row: x[E], x[E + 1]) and rebuilds the 16 factors: 15 products per row and thread.
    const bool rn_rowfac = RN == 1 && LN && cgrp > 1 && !pd.is_last;
    if (RN != 0) {
      const uint32_t ccol = RN == 2 ? cA : cB;                   // the thread's logical column of the tile
      const uint32_t brow = RN == 2 ? gA : baseT;                // its first output row in the round that stores
      const uint64_t colg = pd.rn_first ? 0u : ((uint64_t)ct * TC + ccol); // low output digits = column (passes q >= 1)
      const uint64_t A = pd.rn_first ? 1u : pd.in_sk;            // A_q: the row index counts in units of A_q
      if (!pd.is_last) {
        const uint64_t outer = pd.rn_first ? ((uint64_t)ct * TC + ccol) : (uint64_t)a; // bitrev of the digits still to come
        const uint32_t nb = (uint32_t)pd.rn_next_bits;
        const uint64_t jn = nb ? (uint64_t)(__brev((uint32_t)(outer & ((1u << nb) - 1u))) >> (32 - nb)) : 0u;
        const uint32_t step = tw_load(jn * A * ((uint64_t)1 << QT) * pd.tw_stride);
        wip[0] = tw_load(jn * (colg + A * (NR == 1 ? 0u : brow)) * pd.tw_stride);
#pragma unroll
        for (int m = 1; m < E; m++)
          wip[m] = S::mul(wip[m - 1], step);
        rn_fac = true;
      } else if (nl.inverse) { // X[K] *= 1/N, or g^-K / N (the coset table is pre-scaled by 1/N), K = column + A_q * row
#pragma unroll
        for (int m = 0; m < E; m++)
          wip[m] = COSET ? cpow(colg + A * ((NR == 1 ? 0u : brow) + ((uint64_t)m << QT))) : nl.ninv_mont;
        rn_fac = true;
      }
    } else if (!DIF) {
      // w_M^(jnext * K), K = k (pass 0) or a + n0*k (pass 1). In pass 1 jnext is the column itself, so
      // jnext * K walks the whole table at random (a 64-byte line per 4-byte twiddle, +25 % traffic on that pass);
      // split as w^(jnext*a) * w^(jnext*n0*k): the first index stays in a 256 KiB prefix of the table, the second
      // takes 2^16 distinct values shared by every block -- both live in L2 / Infinity Cache.
      const uint64_t jnext = ((uint64_t)ct * TC + cB) / pd.cprime;
      const bool skip_wa = pd.pidx == 0 || agrp > 1; // (agrp > 1: w^(jnext a) differs per launch row, applied to the operands instead)
      const uint32_t wa = skip_wa ? 0u : tw_load(jnext * a * pd.tw_stride);
#if defined(NTT_WIP_LOADS) // (rounds 1-3, kept for A/B builds: every factor loaded from the table)
#pragma unroll
      for (int m = 0; m < E; m++) {
        const uint32_t k = (NR == 1) ? (uint32_t)m : baseT + ((uint32_t)m << QT);
        if (pd.pidx == 0)
          wip[m] = tw_load(jnext * k * pd.tw_stride);
        else
          wip[m] = S::mul(wa, tw_load(jnext * pd.n0 * k * pd.tw_stride));
      }
#else
      // The thread's E rows k = baseT + (m << QT) form an arithmetic progression, so its factors form a geometric one:
      // two table reads and E - 1 products instead of E reads. In pass 1 those reads are 4-byte gathers, a 64-byte sector each
      // and 8192 of them per block: with few batch rows per block to spread them over (small batches, the lane-native
      // tiles of columns_batch) they were MORE traffic than the data -- 512 KB against 64 KB per row at 2^24 x 8.
      // Exact: table entries are exact powers, so the products are the same field elements, stored canonically.
      const uint64_t kfac = (pd.pidx == 0) ? 1u : (uint64_t)pd.n0;
      const uint32_t step = tw_load(jnext * kfac * ((uint64_t)1 << QT) * pd.tw_stride);
      const uint32_t first = tw_load(jnext * kfac * baseT * pd.tw_stride);
      wip[0] = skip_wa ? first : S::mul(wa, first);
#pragma unroll
      for (int m = 1; m < E; m++)
        wip[m] = S::mul(wip[m - 1], step);
#endif
    }

    // ---- coset factors, once per block (they do not depend on the batch row) ------------------------
    // forward, column pass 0: g^j = g^(column) * g^(row * in_sk); the column part is linear through the whole
    //   column transform and is folded into the inter-pass twiddle, the row part is one product per operand.
    // inverse, last pass: X[K] *= g^-K / N replaces the plain 1/N factor (the table is pre-scaled by 1/N).
    uint32_t cfac[COSET ? E : 1];
    if (!DIF && coset_in) {
      const uint32_t gcol = cpow(in_base + (uint64_t)cB * pd.in_st);
#pragma unroll
      for (int m = 0; m < E; m++)
        wip[m] = S::mul(wip[m], gcol);
#pragma unroll
      for (int u = 0; u < G0; u++) {
        const uint32_t gi = gB * G0 + u;
        const uint32_t kb = (KB_BITS == 0) ? 0u : (__brev(gi) >> (32 - (KB_BITS > 0 ? KB_BITS : 1)));
#pragma unroll
        for (int m = 0; m < (1 << NQ0); m++)
          cfac[u * (1 << NQ0) + m] = cpow(((uint64_t)kb + ((uint64_t)brev_c<NQ0>(m) << KB_BITS)) * pd.in_sk);
      }
    }

    // ---- per-thread HBM offsets (in elements, times the element stride) ----------------------------
    const uint64_t es = nl.es;                            // element stride of what this pass READS
    const uint64_t eso = nl.es_out ? nl.es_out : nl.es;   // ... and of what it WRITES (a work buffer with a padded lane count, ntt_plan.h)
    // column pass: slot (k, t) at in_base + k*sk + t*st on both sides
    // row pass   : load  (k, t) at in_base + k*1  + t*st ; store K0(t) + k_out*out_sk
    const uint64_t K0 = (pd.pidx <= 1) ? ((uint64_t)ct * TC + cB) : (((uint64_t)ct * TC + cB) + (uint64_t)pd.n0 * a);
    const uint64_t K0rev = (DIF && OUTREV) ? bitrev64(K0, nl.logn - SS) : 0; // kNR: see the last round's store
    if (DIF && INV && coset_out) { // slot (u, m) of the lowest round holds X[K0 + (kb + brev(m) * 2^(s-NQ0)) * out_sk]
#pragma unroll
      for (int u = 0; u < G0; u++) {
        const uint32_t gi = gB * G0 + u;
        const uint32_t kb = (NR == 1 || KB_BITS == 0) ? 0u : (__brev(gi) >> (32 - (KB_BITS > 0 ? KB_BITS : 1)));
#pragma unroll
        for (int m = 0; m < (1 << NQ0); m++)
          cfac[u * (1 << NQ0) + m] = cpow(K0 + ((uint64_t)kb + ((uint64_t)brev_c<NQ0>(m) << (NR == 1 ? 0 : KB_BITS))) * pd.out_sk);
      }
    }

    const uint32_t rloc0 = blockIdx.y * rows_per_block;
    auto row_offset = [&](uint32_t rloc, bool rel, bool dst_side = false) -> uint64_t {
      const uint32_t r = rel ? rloc : nl.row0 + rloc; // row inside this launch's group / absolute row
      if (LN) { // r = ((row group * slices) + slice) * cgrp + column of the group
        const uint32_t cs = r % cgrp, rs = r / cgrp;
        return (uint64_t)(rs / nl.lanes) * nl.bs + ((uint64_t)(rs % nl.lanes) << lsh) + (uint64_t)cs * (dst_side ? nl.cst_out : nl.cst_in);
      }
      return (uint64_t)(r / nl.lanes) * nl.bs + (r % nl.lanes);
    };
    // LN: lanes of this row's slice that exist (a partial last slice when ltot is not a multiple of TL)
    auto lane_limit = [&](uint32_t rloc) -> uint32_t { return nl.ltot - ((((nl.row0 + rloc) / cgrp) % nl.lanes) << lsh); };
    // The E operands a thread feeds into its first round, straight from HBM. They are fetched one batch row
    // AHEAD (software prefetch into registers): without it a block alternates between a load phase and a
    // compute/LDS/store phase and the waves sit parked on s_waitcnt for more than half of their cycles
    // (SQ_WAIT_ANY 0.52-0.57 on the column passes, profiles/r01_notes.md).
    auto load_row = [&](uint32_t rloc, uint32_t* x) {
      const uint32_t* __restrict__ pin = in + row_offset(rloc, nl.src_rel != 0);
      if (LN) { // + lane, clamped into the slice: the surplus lanes of a partial slice re-read its last transform
        const uint32_t lim = lane_limit(rloc) - 1u;
        pin += std::min<uint32_t>((DIF && NR > 1) ? lA : lB, lim);
        if (RN == 1) {
          if (rn_rowfac) { // this row's logical column = the block's first + cs * tcl
            const uint64_t ccol = (uint64_t)ct * TC + (uint64_t)((nl.row0 + rloc) % cgrp) * nl.tcl + cB;
            const uint64_t outer = pd.rn_first ? ccol : (uint64_t)a;
            const uint32_t nb = (uint32_t)pd.rn_next_bits;
            const uint64_t jn = nb ? (uint64_t)(__brev((uint32_t)(outer & ((1u << nb) - 1u))) >> (32 - nb)) : 0u;
            const uint64_t A = pd.rn_first ? 1u : pd.in_sk;
            x[E] = tw_load(jn * ((pd.rn_first ? 0u : ccol) + A * (NR == 1 ? 0u : baseT)) * pd.tw_stride);
            x[E + 1] = tw_load(jn * A * ((uint64_t)1 << QT) * pd.tw_stride);
          }
        } else if (!DIF) // middle pass with outer-index groups: this row's w^(jnext (a + cs)), fetched with the operands
          x[E] = agrp > 1 ? tw_load((((uint64_t)ct * TC + cB) / pd.cprime) * (uint64_t)(a + (nl.row0 + rloc) % cgrp) * pd.tw_stride) : 0u;
      }
      if (RN == 2) { // run pass: slots E * gA .. + E - 1 of run cA, as they lie (16 consecutive words of this lane)
        const uint32_t* p = pin + (in_base + (uint64_t)cA * pd.in_st + (uint64_t)E * gA) * es;
        if (E == 16 && nl.vec4) {
#pragma unroll
          for (int g = 0; g < E / 4; g++) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[g];
            x[4 * g] = v.x, x[4 * g + 1] = v.y, x[4 * g + 2] = v.z, x[4 * g + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int m = 0; m < E; m++)
            x[m] = p[(uint64_t)m * es];
        }
      } else if (RN == 1) { // column pass on rows in bit-reversed order: slot (gi << NQ0) + m <- row (gi << NQ0) + m, no reversal
#pragma unroll
        for (int u = 0; u < G0; u++) {
          const uint32_t gi = gB * G0 + u;
          const uint32_t* p = pin + (in_base + ((uint64_t)gi << NQ0) * pd.in_sk + (uint64_t)cB * pd.in_st) * es;
#pragma unroll
          for (int m = 0; m < (1 << NQ0); m++)
            x[u * (1 << NQ0) + m] = p[(uint64_t)m * pd.in_sk * es];
        }
      } else if (V4) { // top round of the row pass: slot 4g+c of the lane in wave-row r <- word c of row 4g+r
        const uint32_t* p = pin + (in_base + (uint64_t)(gA - wrow) + (uint64_t)tA * pd.in_st);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const uint4 v = *reinterpret_cast<const uint4*>(p + ((uint64_t)(4 * g + wrow) << QT));
          x[4 * g] = v.x, x[4 * g + 1] = v.y, x[4 * g + 2] = v.z, x[4 * g + 3] = v.w;
        }
      } else if (!DIF) {
#pragma unroll
        for (int u = 0; u < G0; u++) {
          const uint32_t gi = gB * G0 + u;
          const uint32_t kb = (KB_BITS == 0) ? 0u : (__brev(gi) >> (32 - (KB_BITS > 0 ? KB_BITS : 1)));
          const uint32_t* p = pin + (in_base + (uint64_t)kb * pd.in_sk + (uint64_t)cB * pd.in_st) * es;
          const uint64_t step = (pd.in_sk << KB_BITS) * es;
#pragma unroll
          for (int m = 0; m < (1 << NQ0); m++) // slot m <- source row kb + brev(m) * 2^(s-NQ0)
            x[u * (1 << NQ0) + m] = p[(uint64_t)brev_c<NQ0>(m) * step];
        }
      } else if (NR == 1) {
        const uint32_t* p = pin + (in_base + (uint64_t)cB * pd.in_st) * es;
#pragma unroll
        for (int m = 0; m < (1 << NQ0); m++)
          x[m] = p[(uint64_t)m * es];
      } else { // top round: mapping A, natural rows k = gA + m * L/16
        const uint32_t* p = pin + (in_base + (uint64_t)gA + (uint64_t)cA * pd.in_st) * es;
#pragma unroll
        for (int m = 0; m < 16; m++)
          x[m] = p[((uint64_t)m << QT) * es];
      }
    };
    constexpr bool PREFETCH = NR <= 2;
    const uint32_t nrows = (rloc0 < nl.nrows_launch) ? std::min<uint32_t>(rows_per_block, nl.nrows_launch - rloc0) : 0u;
    // one batch row: rounds, LDS exchanges, stores. `xin` = the E operands of the first round (already in registers).
    // Row rr uses LDS buffer rr & 1; the barrier inside the next row's processing orders the reuse after that.
    auto process_row = [&](uint32_t rr, const uint32_t* xin) {
      const uint32_t rloc = rloc0 + rr;
      uint32_t* __restrict__ pout = out + row_offset(rloc, nl.dst_rel != 0, true) + (LN ? lB : 0u);
      const bool live = !LN || lB < lane_limit(rloc); // (every global store below is in mapping B)
      uint32_t* tile = lds + (size_t)(rr & 1) * (RN == 2 ? T * (L + (L >> 4)) : L * TP);
      if (RN == 2) {
        // ================= run pass: DIT on a contiguous run, lanes along the run =================
        // LDS: column-major, slot sl of run c at c * LC + sl + (sl >> 4) -- a thread's 16 consecutive slots (lowest round) and
        // its 16 slots 2^q0 apart (later rounds) are both conflict-free for the 32 lanes of a half-wave
        constexpr uint32_t LC = L + (L >> 4);
        uint32_t* colp = tile + cA * LC;
        uint32_t x[E];
#pragma unroll
        for (int m = 0; m < E; m++)
          x[m] = xin[m];
#pragma unroll
        for (int u = 0; u < G0; u++)
          ntt_stages<S, NQ0, false, true>(x + u * (1 << NQ0), w0);
        if (NR == 1) {
          uint32_t* q = pout + (in_base + (uint64_t)cA * pd.in_st + (uint64_t)E * gA) * eso;
#pragma unroll
          for (int m = 0; m < E; m++)
            q[(uint64_t)m * eso] = rn_fac ? S::mul(x[m], wip[m]) : x[m];
        } else {
#pragma unroll
          for (int m = 0; m < E; m++)
            colp[(E + 1) * gA + m] = x[m]; // slot E gA + m, + one pad word per 16 slots
          __syncthreads();
#pragma unroll
          for (int r = 1; r < NR; r++) {
            const int q0 = NQ0 + 4 * (r - 1);
            const uint32_t base = ((gA >> q0) << (q0 + 4)) | (gA & ((1u << q0) - 1));
            uint32_t y[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
              const uint32_t sl = base + ((uint32_t)m << q0);
              y[m] = colp[sl + (sl >> 4)];
            }
            ntt_stages<S, 4, false, false>(y, wr[r - 1]);
            if (r == NR - 1) { // slots gA + m * L/16: 4-byte lanes along the run
              uint32_t* q = pout + (in_base + (uint64_t)cA * pd.in_st + (uint64_t)base) * eso;
#pragma unroll
              for (int m = 0; m < 16; m++)
                q[((uint64_t)m << q0) * eso] = rn_fac ? S::mul(y[m], wip[m]) : y[m];
            } else {
#pragma unroll
              for (int m = 0; m < 16; m++) {
                const uint32_t sl = base + ((uint32_t)m << q0);
                colp[sl + (sl >> 4)] = y[m];
              }
              __syncthreads();
            }
          }
        }
      } else if (!DIF) {
        // ================= column pass, DIT =================
        if (RN == 1 && LN) {
          if (rn_rowfac) { // (uniform over the block) this row's own factors, see the note at rn_rowfac
            wip[0] = xin[E];
#pragma unroll
            for (int m = 1; m < E; m++)
              wip[m] = S::mul(wip[m - 1], xin[E + 1]);
          }
        }
#pragma unroll
        for (int u = 0; u < G0; u++) {
          const uint32_t gi = gB * G0 + u;
          uint32_t x[1 << NQ0];
#pragma unroll
          for (int m = 0; m < (1 << NQ0); m++)
            x[m] = xin[u * (1 << NQ0) + m];
          if (LN && agrp > 1) { // (the transform is linear: scaling its operands scales its results)
#pragma unroll
            for (int m = 0; m < (1 << NQ0); m++)
              x[m] = S::mul(x[m], xin[E]);
          }
          if (coset_in) { // row part of g^j (the column part sits in wip)
#pragma unroll
            for (int m = 0; m < (1 << NQ0); m++)
              x[m] = S::mul(x[m], cfac[u * (1 << NQ0) + m]);
          }
          ntt_stages<S, NQ0, false, true>(x, w0);
          if (NR == 1) {
            uint32_t* q = pout + (in_base + (uint64_t)cB * pd.in_st) * eso;
            if (live) {
#pragma unroll
              for (int m = 0; m < (1 << NQ0); m++)
                q[(uint64_t)m * pd.in_sk * eso] = (RN != 0 && !rn_fac) ? x[m] : S::mul(x[m], wip[m]);
            }
          } else {
#pragma unroll
            for (int m = 0; m < (1 << NQ0); m++)
              tile[((gi << NQ0) + m) * TP + tB] = x[m];
          }
        }
        if (NR > 1) {
          __syncthreads();
#pragma unroll
          for (int r = 1; r < NR; r++) {
            const int q0 = NQ0 + 4 * (r - 1);
            const uint32_t base = ((gB >> q0) << (q0 + 4)) | (gB & ((1u << q0) - 1));
            uint32_t x[16];
#pragma unroll
            for (int m = 0; m < 16; m++)
              x[m] = tile[(base + ((uint32_t)m << q0)) * TP + tB];
            ntt_stages<S, 4, false, false>(x, wr[r - 1]);
            if (r == NR - 1) {
              uint32_t* q = pout + (in_base + (uint64_t)base * pd.in_sk + (uint64_t)cB * pd.in_st) * eso;
              const uint64_t step = (pd.in_sk << q0) * eso;
              if (live) {
#pragma unroll
                for (int m = 0; m < 16; m++)
                  q[(uint64_t)m * step] = (RN != 0 && !rn_fac) ? x[m] : S::mul(x[m], wip[m]);
              }
            } else {
#pragma unroll
              for (int m = 0; m < 16; m++)
                tile[(base + ((uint32_t)m << q0)) * TP + tB] = x[m];
              __syncthreads();
            }
          }
        }
      } else {
        // ================= row pass (last pass), DIF =================
        if (NR == 1) {
          uint32_t x[1 << NQ0];
#pragma unroll
          for (int m = 0; m < (1 << NQ0); m++)
            x[m] = xin[m];
          if (coset_in) {
            const uint64_t j0 = in_base + (uint64_t)cB * pd.in_st;
#pragma unroll
            for (int m = 0; m < (1 << NQ0); m++)
              x[m] = S::mul(x[m], cpow(j0 + (uint64_t)m));
          }
          ntt_stages<S, NQ0, true, true>(x, w0);
          if (INV) { // slot m holds X[K0 + brev(m) * out_sk]
#pragma unroll
            for (int m = 0; m < (1 << NQ0); m++)
              x[m] = S::mul(x[m], coset_out ? cfac[m] : nl.ninv_mont);
          }
          if (OUTREV) {
            // (lane-native column groups: launch row cs of the block holds logical column K0 + cs * tcl, and the column enters the bit
            //  reversal -- the three-round store below has had this since round 5, this single-round store did not: 2^9-point
            //  transforms (5 + 4 stages) of 2 or 4 interleaved transforms with kNR / kRR were wrong until the reference's own
            //  randomised test drew one in round 6, tests/test_gpu_ntt_refspace.py)
            const uint64_t k0r = LN ? bitrev64(K0 + (uint64_t)((nl.row0 + rloc) % cgrp) * nl.tcl, nl.logn - SS) : K0rev;
            uint32_t* q = pout + k0r * L * eso;
            if (live) {
#pragma unroll
              for (int m = 0; m < (1 << NQ0); m++)
                q[(uint64_t)m * eso] = x[m];
            }
          } else {
            uint32_t* q = pout + K0 * eso;
            const uint64_t step = pd.out_sk * eso;
            if (live) {
#pragma unroll
              for (int m = 0; m < (1 << NQ0); m++)
                q[(uint64_t)brev_c<NQ0>(m) * step] = x[m];
            }
          }
        } else {
          { // top round: mapping A, natural rows k = gA + m * L/16 (prefetched from HBM)
            uint32_t x[16];
#pragma unroll
            for (int m = 0; m < 16; m++)
              x[m] = xin[m];
            if (V4) {
#pragma unroll
              for (int g = 0; g < 4; g++)
                rows_transpose(x + 4 * g);
            }
            if (coset_in) { // single-pass transform: rows k = gA + m * L/16 of column tA
              const uint64_t j0 = in_base + (uint64_t)gA + (uint64_t)cA * pd.in_st;
#pragma unroll
              for (int m = 0; m < 16; m++)
                x[m] = S::mul(x[m], cpow(j0 + ((uint64_t)m << QT)));
            }
            ntt_stages<S, 4, true, false>(x, wr[NR - 2]);
#pragma unroll
            for (int m = 0; m < 16; m++)
              tile[(gA + ((uint32_t)m << QT)) * TP + tA] = x[m];
          }
          __syncthreads();
#pragma unroll
          for (int r = NR - 2; r >= 1; r--) {
            const int q0 = NQ0 + 4 * (r - 1);
            const uint32_t base = ((gB >> q0) << (q0 + 4)) | (gB & ((1u << q0) - 1));
            uint32_t x[16];
#pragma unroll
            for (int m = 0; m < 16; m++)
              x[m] = tile[(base + ((uint32_t)m << q0)) * TP + tB];
            ntt_stages<S, 4, true, false>(x, wr[r - 1]);
#pragma unroll
            for (int m = 0; m < 16; m++)
              tile[(base + ((uint32_t)m << q0)) * TP + tB] = x[m];
            __syncthreads();
          }
          // lowest round: slot m of group gi holds X[kb + brev(m) * 2^(s-NQ0)], kb = brev(gi)
#pragma unroll
          for (int u = 0; u < G0; u++) {
            const uint32_t gi = gB * G0 + u;
            const uint32_t kb = (KB_BITS == 0) ? 0u : (__brev(gi) >> (32 - (KB_BITS > 0 ? KB_BITS : 1)));
            uint32_t x[1 << NQ0];
#pragma unroll
            for (int m = 0; m < (1 << NQ0); m++)
              x[m] = tile[((gi << NQ0) + m) * TP + tB];
            ntt_stages<S, NQ0, true, true>(x, w0);
            if (INV) { // slot m holds X[K0 + (kb + brev(m) * 2^(s-NQ0)) * out_sk]
#pragma unroll
              for (int m = 0; m < (1 << NQ0); m++)
                x[m] = S::mul(x[m], coset_out ? cfac[u * (1 << NQ0) + m] : nl.ninv_mont);
            }
            if (OUTREV) {
              // bit-reversed output (kNR): bitrev(K0 + k*out_sk) = bitrev_s(k) + L * bitrev(K0), and bitrev_s(k) of
              // slot m is the LDS row (gi << NQ0) + m: a column's L results form one contiguous run of memory.
              // Back into the tile (same slots this thread just read), then stored with lanes along the run.
#pragma unroll
              for (int m = 0; m < (1 << NQ0); m++)
                tile[((gi << NQ0) + m) * TP + tB] = x[m];
            } else if (V4) { // the lane in wave-row r stores slots 4g+r (rows brev4(4g+r)), four adjacent output columns each
              uint32_t* q = pout + ((K0 - wrow) + (uint64_t)kb * pd.out_sk);
              const uint64_t step = pd.out_sk << KB_BITS;
#pragma unroll
              for (int g = 0; g < 4; g++) {
                rows_transpose(x + 4 * g);
                *reinterpret_cast<uint4*>(q + (uint64_t)((wrev << 2) | brev_c<2>(g)) * step) = make_uint4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
              }
            } else {
              uint32_t* q = pout + (K0 + (uint64_t)kb * pd.out_sk) * eso;
              const uint64_t step = (pd.out_sk << KB_BITS) * eso;
              if (live) {
#pragma unroll
                for (int m = 0; m < (1 << NQ0); m++)
                  q[(uint64_t)brev_c<NQ0>(m) * step] = x[m];
              }
            }
          }
          if (OUTREV) {
            __syncthreads();
            // element e = it * nthreads + tid -> (t, r) = (e / L, e % L). With T >= 16: r is fixed per thread and t
            // advances by nthreads / L = T / 16 per step. The column index t sits in the low log2(T) bits of K0,
            // i.e. in the TOP bits of bitrev(K0): bitrev(K0) = bitrev(K0 with t = 0) + (bitrev_lt(t) << (bits - lt)).
            const uint32_t bits = nl.logn - SS, lt = 31 - __clz(T);
            const uint64_t k0b = (pd.pidx <= 1) ? ((uint64_t)ct * TC) : (((uint64_t)ct * TC) + (uint64_t)pd.n0 * a);
            if (LN) { // word e of the tile = (lane, row r, logical column c), lanes fastest: runs of TL words
              // (column groups: launch row cs of the block holds logical columns k0b + cs * tcl + c; the column enters the bit
              //  reversal, so it cannot be a pointer offset: the host sets cst_out = 0 for this store)
              const uint32_t nthr = T * NG16, lim = lane_limit(rloc);
              uint32_t* po = out + row_offset(rloc, nl.dst_rel != 0, true);
              const uint64_t kcol = k0b + (uint64_t)((nl.row0 + rloc) % cgrp) * nl.tcl;
#pragma unroll
              for (int it = 0; it < E; it++) {
                const uint32_t e = (uint32_t)it * nthr + threadIdx.x;
                const uint32_t lane = e & lmask, r = (e >> lsh) & (L - 1u), c = e >> (lsh + SS);
                if (lane < lim) po[(bitrev64(kcol + c, bits) * L + r) * eso + lane] = tile[r * TP + ((c << lsh) | lane)];
              }
            } else if (T >= 16) {
              const uint32_t r = threadIdx.x % L, th = threadIdx.x / L;
              uint32_t* q = pout + (bitrev64(k0b, bits) * L + r) * eso;
#pragma unroll
              for (int it = 0; it < E; it++) {
                const uint32_t t = (uint32_t)it * (T >> 4) + th;
                const uint64_t trev = (uint64_t)(__brev(t) >> (32 - lt));
                q[((trev << (bits - lt)) * L) * eso] = tile[r * TP + t];
              }
            } else { // narrow tiles (tiny first factor): plain per-element addressing
              const uint32_t nthr = T * NG16;
#pragma unroll
              for (int it = 0; it < E; it++) {
                const uint32_t e = (uint32_t)it * nthr + threadIdx.x;
                const uint32_t t = e / L, r = e % L;
                pout[(bitrev64(k0b + t, bits) * L + r) * eso] = tile[r * TP + t];
              }
            }
          }
        }
      }
    };
    if (PREFETCH) {
      // Software prefetch: the next row's E operands are fetched into xnext while this row is processed (without it a
      // block alternates between a load phase and a compute / LDS / store phase and the waves sit parked on s_waitcnt
      // for more than half of their cycles, profiles/r01_notes.md). Two details keep it a prefetch in the ISA:
      //  * the first row must have LANDED before the loop is entered. With "xin pending" on the entry edge and "xin
      //    copied from xnext" on the back edge the waitcnt pass merges the two states conservatively and guards the
      //    first butterflies of EVERY iteration with vmcnt(6..0) right after the next row's loads were issued: it waits
      //    for those very loads (that is what rounds 1-2 shipped, profiles/r02_notes.md);
      //  * the copy xin <- xnext stays behind the stores (sched_barrier): hoisted into the second round it needs the
      //    loads half an iteration early.
      constexpr int EX = E + (LN ? (RN == 1 ? 2 : 1) : 0); // LN: one (RN: two) more words per row (see load_row)
      uint32_t xin[EX], xnext[EX];
      if (nrows) {
        load_row(rloc0, xin);
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
      }
      for (uint32_t rr = 0; rr < nrows; rr++) {
        const bool has_next = rr + 1 < nrows;
        if (has_next) load_row(rloc0 + rr + 1, xnext);
        process_row(rr, xin);
        if (has_next) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < EX; m++)
            xin[m] = xnext[m];
        }
      }
    } else { // (three-round variants, s >= 9, are already at 110-150 VGPRs: they load each row when it is needed)
      for (uint32_t rr = 0; rr < nrows; rr++) {
        uint32_t xin[E + (LN ? (RN == 1 ? 2 : 1) : 0)];
        load_row(rloc0 + rr, xin);
        process_row(rr, xin);
      }
    }
  }

  using pass_fn_t = void (*)(const uint32_t*, uint32_t*, const uint32_t*, const uint32_t*, PassDesc, NttLaunch, uint32_t);

  template <class PR, bool LN, int NQ0, int NR, bool COSET>
  static pass_fn_t pick_variant2(bool dif, bool inv, bool outrev)
  {
    if (!dif) return (pass_fn_t)k_ntt_fast<PR, NQ0, NR, false, false, COSET, false, false, false, LN>;
    if (outrev)
      return inv ? (pass_fn_t)k_ntt_fast<PR, NQ0, NR, true, true, COSET, true, false, false, LN> : (pass_fn_t)k_ntt_fast<PR, NQ0, NR, true, false, COSET, true, false, false, LN>;
    return inv ? (pass_fn_t)k_ntt_fast<PR, NQ0, NR, true, true, COSET, false, false, false, LN> : (pass_fn_t)k_ntt_fast<PR, NQ0, NR, true, false, COSET, false, false, false, LN>;
  }
  template <class PR, bool LN, int NQ0, int NR>
  static pass_fn_t pick_variant(bool dif, bool inv, bool coset, bool outrev)
  {
    return coset ? pick_variant2<PR, LN, NQ0, NR, true>(dif, inv, outrev) : pick_variant2<PR, LN, NQ0, NR, false>(dif, inv, outrev);
  }
  // the 4-byte-lane variant for a pass of 2^s points; instantiates all of them for <PR, LN>
  template <class PR, bool LN>
  static pass_fn_t pick_pass_t(int s, bool dif, bool inv, bool coset, bool outrev)
  {
    switch (s) {
    case 1: return pick_variant<PR, LN, 1, 1>(dif, inv, coset, outrev);
    case 2: return pick_variant<PR, LN, 2, 1>(dif, inv, coset, outrev);
    case 3: return pick_variant<PR, LN, 3, 1>(dif, inv, coset, outrev);
    case 4: return pick_variant<PR, LN, 4, 1>(dif, inv, coset, outrev);
    case 5: return pick_variant<PR, LN, 1, 2>(dif, inv, coset, outrev);
    case 6: return pick_variant<PR, LN, 2, 2>(dif, inv, coset, outrev);
    case 7: return pick_variant<PR, LN, 3, 2>(dif, inv, coset, outrev);
    case 8: return pick_variant<PR, LN, 4, 2>(dif, inv, coset, outrev);
    case 9: return pick_variant<PR, LN, 1, 3>(dif, inv, coset, outrev);
    case 10: return pick_variant<PR, LN, 2, 3>(dif, inv, coset, outrev);
    case 11: return pick_variant<PR, LN, 3, 3>(dif, inv, coset, outrev);
    case 12: return pick_variant<PR, LN, 4, 3>(dif, inv, coset, outrev);
    }
    return nullptr;
  }

  // bit-reversed input consumed natively: mode 2 = run pass (pass 0 of a row-major batch), mode 1 = column-type pass
  template <class PR, bool LN, int NQ0, int NR>
  static pass_fn_t pick_variant_rn(int mode, bool coset)
  {
    if (mode == 2) {
      if constexpr (!LN)
        return coset ? (pass_fn_t)k_ntt_fast<PR, NQ0, NR, false, false, true, false, false, false, false, 2> : (pass_fn_t)k_ntt_fast<PR, NQ0, NR, false, false, false, false, false, false, false, 2>;
      else
        return nullptr;
    }
    return coset ? (pass_fn_t)k_ntt_fast<PR, NQ0, NR, false, false, true, false, false, false, LN, 1> : (pass_fn_t)k_ntt_fast<PR, NQ0, NR, false, false, false, false, false, false, LN, 1>;
  }
  template <class PR, bool LN>
  static pass_fn_t pick_pass_rn_t(int s, int mode, bool coset)
  {
    switch (s) {
    case 1: return pick_variant_rn<PR, LN, 1, 1>(mode, coset);
    case 2: return pick_variant_rn<PR, LN, 2, 1>(mode, coset);
    case 3: return pick_variant_rn<PR, LN, 3, 1>(mode, coset);
    case 4: return pick_variant_rn<PR, LN, 4, 1>(mode, coset);
    case 5: return pick_variant_rn<PR, LN, 1, 2>(mode, coset);
    case 6: return pick_variant_rn<PR, LN, 2, 2>(mode, coset);
    case 7: return pick_variant_rn<PR, LN, 3, 2>(mode, coset);
    case 8: return pick_variant_rn<PR, LN, 4, 2>(mode, coset);
    case 9: return pick_variant_rn<PR, LN, 1, 3>(mode, coset);
    case 10: return pick_variant_rn<PR, LN, 2, 3>(mode, coset);
    case 11: return pick_variant_rn<PR, LN, 3, 3>(mode, coset);
    case 12: return pick_variant_rn<PR, LN, 4, 3>(mode, coset);
    }
    return nullptr;
  }
  pass_fn_t pick_pass_rn_lanes_babybear(int s, bool coset);
  pass_fn_t pick_pass_rn_lanes_koalabear(int s, bool coset);
  template <class PR>
  static pass_fn_t pick_pass_rn_lanes(int s, bool coset)
  {
    return PR::P == babybear_params::P ? pick_pass_rn_lanes_babybear(s, coset) : pick_pass_rn_lanes_koalabear(s, coset);
  }

  // lane-native variants (LN = true), compiled in ntt_lanes.hip
  pass_fn_t pick_pass_lanes_babybear(int s, bool dif, bool inv, bool coset, bool outrev);
  pass_fn_t pick_pass_lanes_koalabear(int s, bool dif, bool inv, bool coset, bool outrev);
  template <class PR>
  static pass_fn_t pick_pass_lanes(int s, bool dif, bool inv, bool coset, bool outrev)
  {
    return PR::P == babybear_params::P ? pick_pass_lanes_babybear(s, dif, inv, coset, outrev) : pick_pass_lanes_koalabear(s, dif, inv, coset, outrev);
  }

} // namespace icicle_hip
