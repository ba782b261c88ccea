// Host-side planning helpers of the MSM that carry no device code (also compiled by tests/plan_harness.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include "../../include/icicle_hip.h"

namespace icicle_hip {

  constexpr int MSM_MAX_GROUPS = 16;

  // Window groups of the pipelined schedule (msm_impl.hpp): `tw` windows cut into at most `want` groups, group 0 holding
  // the HIGHEST windows. Group g = [glo[g], ghi[g]); the end groups are half as wide as the inner ones (a short first
  // group lets accumulation start early, a short last one leaves little to reduce at the end). Returns the group count.
  static inline int msm_window_groups(int tw, int want, int* glo, int* ghi)
  {
    int NG = std::max(1, std::min(std::min(want, MSM_MAX_GROUPS), tw / 2));
    if (tw < 4) NG = 1;
    const double unit = (double)tw / (NG <= 2 ? NG : NG - 1);
    double acc = 0;
    int hi = tw;
    for (int g = 0; g < NG; g++) {
      acc += (NG <= 2 || (g > 0 && g < NG - 1)) ? unit : unit / 2;
      int lo = g == NG - 1 ? 0 : std::max(0, tw - (int)(acc + 0.5));
      lo = std::min(lo, hi - 1);
      if (g < NG - 1) lo = std::max(lo, NG - 1 - g); // every later group keeps at least one window
      glo[g] = lo, ghi[g] = hi;
      hi = lo;
    }
    return NG;
  }

  #ifndef MSM_DEFAULT_GROUPS
  #define MSM_DEFAULT_GROUPS 1 // window groups of the pipelined schedule (msm_run_single); ICICLE_HIP_MSM_GROUPS overrides
  #endif
  struct MsmPlan {
    int bits;    // scalar bits considered
    int c;       // window bits (of the widest windows)
    int nwin;    // total windows W = ceil((bits+1)/c)   [mixed plan: chosen W, sum of the widths = bits]
    int pf;      // precompute factor
    int wpf;     // windows per precomputed base = target windows actually accumulated
    uint32_t nb; // buckets per window = 2^(c-1) (array stride of every per-bucket table)
    uint32_t seg; // bucket-accumulation segment size: a bucket with more points is split across threads
    // Mixed window widths (round 5): windows [0, n_lo) are c - 1 bits wide and use the first nb / 2 buckets of their slot,
    // windows [n_lo, nwin) are c bits wide; n_lo = 0 is the uniform plan of rounds 1-4. With `negate` a scalar whose top bit
    // is set is replaced by r - s and its point negated (the reference's own trick, cpu_msm.hpp:276-277), so scalars stay
    // below 2^(bits-1) and the widths only have to add up to `bits`: 254 bits = 10 x 21 + 2 x 22, twelve windows instead of
    // the thirteen of c = 20 (-7.7 % mixed additions) for 2.15 x the buckets (uniform c = 22: 3.7 x, measured a wash).
    int n_lo = 0;
    bool negate = false;
    int width(int w) const { return w < n_lo ? c - 1 : c; }
    int offset(int w) const { return w < n_lo ? w * (c - 1) : n_lo * (c - 1) + (w - n_lo) * c; } // bit offset of window w
    uint32_t nb_of(int w) const { return w < n_lo ? nb / 2 : nb; }
    size_t buckets_used() const { return (size_t)n_lo * (nb / 2) + (size_t)(nwin - n_lo) * nb; }
  };

  // force_windows: 0 = the cost model decides; W > 0 = W windows whose widths add up to the scalar bits (tests, A/B:
  // MSMConfig.ext "hip_msm_windows"); -1 = uniform widths only (callers that fix config.c for several sub-calls)
  static MsmPlan make_plan(int n, int scalar_bits, const icicle_msm_config_t& cfg, int force_windows = 0)
  {
    MsmPlan p;
    p.bits = (cfg.bitsize > 0 && cfg.bitsize < scalar_bits) ? cfg.bitsize : scalar_bits;
    p.pf = std::max(1, cfg.precompute_factor);
    int c = cfg.c;
    // A precomputed base table fixes the doubling shift c * wpf, so msm_precompute_bases(nof_bases) and msm(msm_size)
    // must agree on c. Like the reference (cpu_msm.hpp:466 vs :207, get_optimal_c :103-119) both sides derive it from the
    // size of ONE MSM, the scalar bits and precompute_factor -- and from nothing else: with precompute_factor > 1 the
    // choice ignores batch_size (the two calls rarely carry the same one: the Rust suite precomputes shared bases with
    // batch_size 1 and then runs batches of 1, 3 and 16 on the table, wrappers/rust/icicle-core/src/msm/tests.rs:96-134).
    // msm_precompute_bases is handed the size of one MSM as nof_bases by both wrappers (msm_impl.hpp msm_precompute_run), and
    // msm() on a table this process wrote takes c from the table registry (common.h) before it gets here. Rounds 1-3 used a fixed c = 16 here, which made every MSM from 2^20 up
    // SLOWER with a table (16 windows against 13); now a table lowers the bucket count per window set by pf and the cost
    // model moves c up with it, as docs/docs/api/cpp/msm.md:155-201 describes. Pass config.c on both calls to override.
    const bool table = p.pf > 1;
    if (c <= 0) {
      // minimise  (#mixed adds) + (bucket-reduction work). Fitted to a measured sweep (profiles/r03_msm_csweep.txt):
      //  * a window size whose TOP window holds only 1-3 scalar bits is never chosen: that window has a handful of
      //    buckets with n/4 points each, i.e. one more window of mixed adds for nothing plus the whole overflow machinery
      //    (c = 14, 18, 21 for 254-bit scalars: 2^20 4.7 / 4.6 ms against 2.9 ms at c = 17; 2^26 98 against 71 ms);
      //  * n >= 2^16: with fewer than ~2.5 waves of bucket threads per SIMD the accumulation runs below its issue rate,
      //    which favours MORE buckets than the add count alone suggests, and a bucket costs 2 (below 2^22) to 4 add-
      //    equivalents in the reduction (2^16: c 13 -> 15, 2.10 -> 1.74 ms; 2^18: 2.67 -> 2.15; 2^20: 15 -> 17, 3.4 -> 2.9);
      //  * below 2^16 the latency of the reduction and of the window combine dominates: per bucket: ~2 complete adds
      //    (14 muls each) vs 10 muls per mixed add, weighted 8 for their poor parallelism (round 1's fit, still the best).
      // (a batch runs as ONE launch sequence with batch x the bucket threads and batch x the reduction work: round 1's
      //  weights stay the better fit there -- 16 x 2^16: 3.5 ms against 4.5 ms with the single-MSM fit)
      const bool mid = n >= (1 << 16) && (table || std::max(1, cfg.batch_size) == 1);
      // A single small MSM of full-width scalars is a chain of latencies, and since the bucket reduction runs four lanes per column
      // (round 6, msm_impl.hpp k_reduce_wave_quad) a bucket costs little: what counts is the length of the accumulation chains
      // (n / 2^(c-1) mixed additions per bucket) and the window count. Measured sweeps (profiles/r06_small_msm_c_sweep.txt): up to 2^12
      // terms c = 8 (32 windows; BN254 2^10 1.23 -> 0.89 ms, BLS12-381 2^11 2.14 -> 1.58), above c = 15 (17 windows of a 254-bit scalar;
      // BN254 2^15 1.48 -> 1.09 ms) -- or the next c whose top window is not a stub (BLS12-381: 16; 2^15 2.74 -> 2.05 ms).
      int small_c = 0;
      // (up to 2^18 terms: from 2^16 the fit below picks 15 for 253 / 254-bit scalars itself, but 13 / 14 for 255-bit ones where 15 leaves a one-bit
      //  top window -- BLS12-381 2^16 2.81 ms against 2.14 at c = 16, 2^17 2.94 against 2.29)
      if (!table && std::max(1, cfg.batch_size) == 1 && n < (1 << 18) && p.bits >= 200) {
        small_c = n <= (1 << 12) ? 8 : 15;
        for (;; small_c++) {
          const int w = (p.bits + 1 + small_c - 1) / small_c;
          if (!(w > 1 && p.bits + 1 - small_c * (w - 1) <= 3)) break;
        }
      }
      double best = 1e300;
      for (int cc = (small_c ? small_c : 2); cc <= (small_c ? small_c : 21); cc++) {
        const int w = (p.bits + 1 + cc - 1) / cc;
        const int wpf = (w + p.pf - 1) / p.pf;
        if (w > 1 && p.bits + 1 - cc * (w - 1) <= 3 && p.bits > 8) continue; // tiny top window
        // batches of small MSMs: the one-level sort (c <= 11) beats the two-level one by far while the windows are small
        // (128 x 2^17: c = 11 29.2 ms, c = 12 / 13 36.4 / 34.8; 1024 x 2^12: c = 12 77 ms against 16) -- profiles/r03_notes.md 11
        if (!table && cfg.batch_size > 1 && n <= (1 << 17) && cc > 11) continue;
        const double nbk = (double)wpf * (double)(1u << (cc - 1));
        double cost;
        if (mid) {
          const double occupancy = std::max(1.0, 2.5 * 65536.0 / nbk); // bucket threads per SIMD lane slot
          // (below 2^22 the reduction is latency-, not throughput-bound -- but with a base table there are few bucket sets, and a
          //  wider window buys them at the full price: fitted to profiles/r04_precompute_csweep.txt, e.g. 2^18, pf 8: c = 17 1.96 ms,
          //  c = 19 2.29 ms)
          cost = (double)w * n * occupancy + ((table || n >= (1 << 22)) ? 4.0 : 2.0) * nbk;
        } else {
          cost = (double)w * n + 8.0 * nbk;
        }
        if (cost < best) {
          best = cost;
          c = cc;
        }
      }
    }
    // two-level sort: 2^hb partitions (pass A, hb <= 10) x 2^lb bins (pass B, lb <= 11). The model above stops at 21: a caller-set
    // 22 runs (12 windows of a 254-bit scalar instead of 13, four times the buckets of c = 20) and was measured slower at 2^26
    // (profiles/r04_msm_csweep.txt)
    c = std::min(22, std::max(2, c));
    p.c = c;
    p.nwin = (p.bits + 1 + c - 1) / c;
    p.wpf = (p.nwin + p.pf - 1) / p.pf;
    p.nb = 1u << (c - 1);
    // ---- mixed widths: W windows, the top x of them one bit wider, widths adding up to the scalar bits exactly. Needs the
    // negation trick, i.e. full-width scalars (a chopped scalar has no top bit to test, cpu_msm.hpp:276), no base table (its
    // doubling shift is c * wpf) and the two-level sort (c - 1 >= 12). The cost model compares it with the best uniform plan
    // at the same weights: it wins from ~2^25 terms up, where a window of mixed additions outweighs twice the buckets.
    if (cfg.c <= 0 && p.pf == 1 && p.bits == scalar_bits && std::max(1, cfg.batch_size) == 1 && (force_windows > 0 || (force_windows == 0 && n >= (1 << 23)))) {
      auto uniform_cost = [&](int w, double nbk) {
        const double occupancy = std::max(1.0, 2.5 * 65536.0 / nbk);
        return (double)w * n * occupancy + 4.0 * nbk;
      };
      int bestW = 0;
      double best = force_windows > 0 ? 1e300 : 0.995 * uniform_cost(p.nwin, (double)p.nwin * p.nb);
      for (int W = (force_windows > 0 ? force_windows : 2); W <= (force_windows > 0 ? force_windows : 40); W++) {
        const int clo = p.bits / W, x = p.bits - clo * W; // x windows of clo + 1 bits, W - x of clo bits
        if (x == 0 || clo + 1 > 22 || clo < (force_windows > 0 ? 2 : 12)) continue;
        const double nbk = (double)(W - x) * (double)(1u << (clo - 1)) + (double)x * (double)(1u << clo);
        const double cost = force_windows > 0 ? 0.0 : uniform_cost(W, nbk);
        if (cost < best) best = cost, bestW = W;
      }
      if (bestW > 0) {
        const int clo = p.bits / bestW, x = p.bits - clo * bestW;
        p.c = clo + 1;
        p.nwin = p.wpf = bestW;
        p.n_lo = bestW - x;
        p.negate = true;
        p.nb = 1u << (p.c - 1);
      }
    }
    {
      // points per bucket: the nwin windows of a scalar land in wpf bucket sets. (Rounds 1-3 had one more factor pf here: with
      // a base table the segments came out pf times too long, and the handful of buckets that take the whole short top window
      // -- n / 2^(top bits - 1) points each -- were walked by a few threads in chains of thousands of additions: pf = 8, c = 19
      // at 2^24 took 49.9 ms against 19.9 ms at c = 20, profiles/r04_precompute_sweep.txt.)
      const double avg = (double)n * ((double)p.nwin / p.wpf) / (double)(p.n_lo > 0 ? p.nb / 2 : p.nb); // (of the narrow windows)
      uint32_t sgm = 64;
      while ((double)sgm < 2.0 * avg)
        sgm <<= 1;
      p.seg = sgm;
    }
    return p;
  }


} // namespace icicle_hip
