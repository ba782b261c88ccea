// Pass decomposition shared by the 31-bit NTT (ntt.hip) and the 256-bit scalar-field NTT (ntt_big.hip).
//
// N = N0*N1[*N2] is split into at most three passes; pass p computes 2^s-point sub-transforms on
// [L x T] tiles (T adjacent columns => every HBM access is a run of T contiguous elements), applies
// the inter-pass twiddle on the way out, and the LAST pass scatters straight into natural order.
// Addresses are in ELEMENTS of one logical row; the kernels scale by the element stride / width.
#pragma once
#include <algorithm>
#include <cstdint>

namespace icicle_hip {

  // ---- pass descriptor -------------------------------------------------------------------------
  struct PassDesc {
    int s;            // log2 of the sub-transform length L
    int T;            // tile width (columns per block)
    uint32_t ntiles;  // tiles per row-transform
    // tile -> (a, c0): tile index = a * tiles_per_a + ct ;
    uint32_t tiles_per_a;
    // load/store addressing inside one logical row: addr = base + k*sk + t*st
    uint64_t in_base_a, in_base_ct, in_sk, in_st;     // base = a*in_base_a + ct*in_base_ct
    uint64_t out_sk, out_st;                          // out base computed in-kernel (needs digit reversal)
    int is_last;      // last pass: natural-order scatter + optional 1/N scaling
    int pidx;         // pass index
    int xcd_remap;    // fast path: blockIdx.x -> tile map that keeps neighbouring tiles on one XCD
    // twiddle after this pass (not last): exponent = jnext * K, table stride tstride
    uint64_t tw_stride;  // max / M
    uint32_t cprime;     // C' = C / N_{p+1}; jnext = (c0 + t) / C'
    uint32_t n0, n1;     // N_0, N_1 (for K and digit reversal)
    // bit-reversed INPUT consumed natively (kRN, ntt_fast.hpp RN != 0): in-place DIT by digits. Pass q transforms digit j_q
    // (rows in bit-reversed order in memory, read as they lie) into k_q (rows in natural order); the columns of pass q >= 1 are
    // the A_q = N_0 .. N_{q-1} low output digits already produced, the outer index `a` is the bit reversal of the digits still
    // to come. rn_first: pass 0, whose rows are the elements of a contiguous run (the tile's columns are runs, not output digits).
    // rn_next_bits = log2 N_{q+1}: the factor behind pass q is w_{M}^(j_{q+1} * (column + A_q * row)), j_{q+1} = bitrev(a mod N_{q+1}).
    int rn_first = 0, rn_next_bits = 0;
  };

  struct NttLaunch {
    uint32_t logn;
    uint64_t n;
    uint32_t nbatch;   // number of independent lane-transforms (batch * lanes)
    uint32_t lanes;    // 1 (scalar) or 4 (quartic extension)
    uint64_t bs;       // offset(b') = (b'/lanes)*bs + (b'%lanes)
    uint64_t es;       // element stride (of the pass's source)
    // element stride of the pass's DESTINATION when it differs (0 = es). Interleaved transforms whose count is not a multiple of
    // 32 (columns_batch with 100 columns: rows of 400 bytes) make every 128-byte access of every pass straddle three 64-byte sectors
    // instead of two (2.9 TB/s per pass against 4.8, profiles/r05_notes.md section 5). Only the passes that touch the CALLER's
    // buffers have to see that layout: the work buffer between the passes pads the lane count to a multiple of 32 words, so of the
    // six memory sides of a three-pass transform four are sector-aligned (the padding lanes are never read or written).
    uint64_t es_out = 0;
    int in_rev, out_rev; // bit-reversed logical->memory maps
    int inverse;
    uint32_t log_max;
    uint32_t ninv_mont;  // N^-1 (Montgomery) for inverse
    int coset;           // multiply by powers table (forward: on first load; inverse: on last store)
    // row-group execution (fast path): this launch covers rows [row0, row0 + nrows_launch); a buffer
    // flagged "relative" holds only the current group (its row r lives at offset of row r - row0)
    uint32_t row0 = 0, nrows_launch = 0;
    int src_rel = 0, dst_rel = 0;
    // lane-native tiles (ntt_fast.hpp, LN): `ltot` transforms interleaved word by word, a launch row = one slice of
    // 2^lsh of them; `lanes` then counts the slices per row group: offset(r) = (r / lanes) * bs + ((r % lanes) << lsh)
    uint32_t lsh = 0, ltot = 1;
    // lane-native, few slices (16-64 interleaved transforms): `cgrp` ADJACENT logical columns share one twiddle set in pass 0
    // (same w_M^(jnext K): jnext = column / cprime) and in the last pass (no inter-pass factor at all), so they run as launch
    // rows of one block: row r = (row group, slice, column cs), cs = r % cgrp, at word offsets cs * cst_in / cs * cst_out
    // In the middle pass of a three-pass transform the factor is w^(column * (a + n0 k)): there `agrp` adjacent values of the OUTER
    // index a run as the rows of a block (agrp = cgrp), w^(column * a) being applied per row to the loaded operands.
    uint32_t cgrp = 1, agrp = 1;
    int vec4 = 0;     // RN run pass: unit element stride and 16-byte aligned rows -- a thread's 16 consecutive words load as 4 x uint4
    uint32_t tcl = 1; // logical columns in the LDS tile (PassDesc::T counts the cgrp-wide group in pass 0 / the last pass)
    uint64_t cst_in = 0, cst_out = 0;
  };

#if defined(__HIPCC__)
  __device__ __forceinline__ uint64_t bitrev64(uint64_t x, uint32_t bits)
  {
    return bits == 0 ? 0 : (__brevll(x) >> (64 - bits));
  }
#endif


  // <= 3 passes; sub-transforms of at most 2^smax points while that covers logn, larger above
  static inline void split_logn(int logn, int smax, int* parts, int* np)
  {
    const int SMAX = std::max(smax, (logn + 2) / 3);
    const int P = std::max(1, (logn + SMAX - 1) / SMAX);
    for (int i = 0; i < P; i++)
      parts[i] = logn / P + (i < logn % P ? 1 : 0);
    *np = P;
  }

  // descriptor of pass p of P; tmax = widest tile the kernel's LDS / block budget allows for 2^parts[p] rows
  static inline PassDesc make_pass(const int* parts, int P, int p, uint64_t n, int log_max, uint32_t tmax)
  {
    PassDesc pd{};
    pd.s = parts[p];
    pd.pidx = p;
    pd.is_last = (p == P - 1);
    const uint64_t L = (uint64_t)1 << pd.s;
    uint64_t A = 1, C = 1;
    for (int q = 0; q < p; q++)
      A <<= parts[q];
    for (int q = p + 1; q < P; q++)
      C <<= parts[q];
    pd.n0 = 1u << parts[0];
    pd.n1 = P > 1 ? (1u << parts[1]) : 1;
    if (!pd.is_last) {
      pd.T = (int)std::min<uint64_t>(tmax, C);
      pd.tiles_per_a = (uint32_t)(C / pd.T);
      pd.ntiles = (uint32_t)(A * pd.tiles_per_a);
      pd.in_base_a = L * C;
      pd.in_base_ct = pd.T;
      pd.in_sk = C;
      pd.in_st = 1;
      int lm = 0; // M = N_0..N_{p+1}
      for (int q = 0; q <= p + 1; q++)
        lm += parts[q];
      pd.tw_stride = (uint64_t)1 << (log_max - lm);
      pd.cprime = (uint32_t)(C >> parts[p + 1]);
    } else {
      // tile over k0 (the slowest digit of a); for P==3 `a` in the kernel carries k1
      const uint64_t n0 = P >= 2 ? ((uint64_t)1 << parts[0]) : 1;
      const uint64_t n1 = P == 3 ? ((uint64_t)1 << parts[1]) : 1;
      pd.T = (int)std::min<uint64_t>(tmax, n0);
      pd.tiles_per_a = (uint32_t)(n0 / pd.T);
      pd.ntiles = (uint32_t)(n1 * pd.tiles_per_a);
      // row index = k0*n1 + k1 ; row start = row*L
      pd.in_base_a = L; // a = k1
      pd.in_base_ct = (uint64_t)pd.T * n1 * L;
      pd.in_sk = 1;
      pd.in_st = n1 * L;
      pd.out_sk = n >> pd.s;
      pd.out_st = 1;
    }
    return pd;
  }

  // Pass q of the native bit-reversed-input pipeline (see PassDesc::rn_first). run_pass: pass 0 of a row-major batch, run by
  // the RN == 2 code (lanes along the run); otherwise pass 0 has the column shape too (lane-native tiles: the tile's word-columns
  // are interleaved transforms, a row = one position of the run).
  static inline PassDesc make_pass_rn(const int* parts, int P, int q, int log_max, uint32_t tmax)
  {
    PassDesc pd{};
    pd.s = parts[q];
    pd.pidx = q;
    pd.is_last = (q == P - 1);
    const uint64_t L = (uint64_t)1 << pd.s;
    uint64_t C = 1, O = 1; // columns = low output digits done; outer = input digits still to come
    for (int i = 0; i < q; i++)
      C <<= parts[i];
    for (int i = q + 1; i < P; i++)
      O <<= parts[i];
    pd.n0 = 1u << parts[0];
    pd.n1 = P > 1 ? (1u << parts[1]) : 1;
    pd.rn_first = (q == 0);
    if (q == 0) { // tile = T runs of L contiguous elements
      pd.T = (int)std::min<uint64_t>(tmax, O);
      pd.tiles_per_a = (uint32_t)(O / pd.T);
      pd.ntiles = pd.tiles_per_a;
      pd.in_base_a = 0;
      pd.in_base_ct = (uint64_t)pd.T * L;
      pd.in_sk = 1;
      pd.in_st = L;
    } else {
      pd.T = (int)std::min<uint64_t>(tmax, C);
      pd.tiles_per_a = (uint32_t)(C / pd.T);
      pd.ntiles = (uint32_t)(O * pd.tiles_per_a);
      pd.in_base_a = L * C;
      pd.in_base_ct = pd.T;
      pd.in_sk = C;
      pd.in_st = 1;
    }
    if (!pd.is_last) {
      int lm = 0; // M = N_0 .. N_{q+1}
      for (int i = 0; i <= q + 1; i++)
        lm += parts[i];
      pd.tw_stride = (uint64_t)1 << (log_max - lm);
      pd.rn_next_bits = parts[q + 1];
    }
    return pd;
  }

  // ---- one transform split over P device slots (ntt_split.hpp): N = 2^a * 2^b, both factors at least P wide ----
  struct SplitShape {
    int P, logn, a, b; // N1 = 2^a, N2 = 2^b
  };
  static inline bool split_shape(int logn, int P, SplitShape* s, bool allow_one = false)
  { // (P = 1: the "hip_force_rccl" rehearsal of the exchanges on a single device slot)
    if (P < (allow_one ? 1 : 2) || (P & (P - 1)) != 0) return false;
    int lp = 0;
    while ((1 << lp) < P)
      lp++;
    const int a = std::max((logn + 1) / 2, lp), b = logn - a;
    if (b < lp || (P == 1 && b < 1)) return false;
    *s = {P, logn, a, b};
    return true;
  }

  // ---- ECNTT stage plan (ecntt.hip): radix-2^r "matrix form" stages ---------------------------------------------------
  // r = the largest value <= 5 whose tot (2^r - 1) / 2 scalar multiplications per stage stay within `budget` quads (one round
  // of quads on the chip); forced_r > 0 overrides. widths[] receives the stage widths (as even as possible, sum = logn),
  // returns their count; r = 1 everywhere means the plain radix-2 kernel.
  static inline int ecntt_stage_plan(int logn, uint64_t tot, uint64_t budget, int forced_r, int* widths)
  {
    int rmax = 1;
    if (forced_r > 0) {
      rmax = std::min(5, forced_r);
    } else {
      while (rmax < 5 && tot * ((2ull << rmax) - 1) / 2 <= budget)
        rmax++;
    }
    rmax = std::max(1, std::min(rmax, std::max(1, logn)));
    const int nst = logn > 0 ? (logn + rmax - 1) / rmax : 0;
    for (int si = 0; si < nst; si++)
      widths[si] = logn / nst + (si < logn % nst ? 1 : 0);
    return nst;
  }
  // exponent (mod M = 2^(q0 + r)) of the twiddle that term (j, u) of a radix-2^r stage at q0 multiplies A_j[pos] with:
  // Y[pos + L k] = sum_j w_M^(rev_r(j) (pos + L k)) A_j[pos], k = u or u + R/2 (the latter with the sign (-1)^rev_r(j))
#if defined(__HIPCC__)
  __host__ __device__
#endif
  static inline uint64_t ecntt_term_exponent(int q0, int r, uint32_t j, uint32_t u, uint64_t pos)
  {
    uint32_t rj = 0;
    for (int b = 0; b < r; b++)
      rj |= ((j >> b) & 1u) << (r - 1 - b);
    const uint64_t L = (uint64_t)1 << q0, Mmask = ((uint64_t)1 << (q0 + r)) - 1;
    return ((uint64_t)rj * (pos + L * u)) & Mmask;
  }

} // namespace icicle_hip
