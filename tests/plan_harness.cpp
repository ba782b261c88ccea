// Host check of the NTT pass plan (icicle_amd/csrc/ntt_plan.h): for every size, every pass's tiles must touch each
// slot of a row exactly once on the way in, and the last pass's natural-order scatter must hit each output once.
// The address formulas are the kernels' (k_ntt_fast / k_ntt_pass_generic / k_big_ntt_pass).
#include "../icicle_amd/csrc/ntt_plan.h"
#include "../icicle_amd/csrc/msm_plan.h"
#include <cstdio>
#include <vector>
using namespace icicle_hip;

static int check(int logn, int smax, uint32_t tmax_cap)
{
  int parts[3], P;
  split_logn(logn, smax, parts, &P);
  int sum = 0;
  for (int i = 0; i < P; i++)
    sum += parts[i];
  if (sum != logn || P < 1 || P > 3) return 1;
  const uint64_t n = 1ull << logn;
  for (int p = 0; p < P; p++) {
    const uint64_t L = 1ull << parts[p];
    const uint64_t epb = L >= 16 ? 16 : L;
    uint32_t tmax = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(tmax_cap, 512 * epb / L));
    while (tmax > 1 && 2 * L * (tmax + 1) * 4 > 160 * 1024)
      tmax >>= 1;
    const PassDesc pd = make_pass(parts, P, p, n, /*log_max=*/logn, tmax);
    if (pd.T < 1 || (uint64_t)pd.ntiles * pd.T * L != n) return 10 + p;
    std::vector<uint8_t> in(n, 0), out(n, 0);
    for (uint32_t tile = 0; tile < pd.ntiles; tile++) {
      const uint32_t a = tile / pd.tiles_per_a, ct = tile % pd.tiles_per_a;
      const uint64_t in_base = (uint64_t)a * pd.in_base_a + (uint64_t)ct * pd.in_base_ct;
      for (uint64_t k = 0; k < L; k++)
        for (int t = 0; t < pd.T; t++) {
          const uint64_t addr = in_base + k * pd.in_sk + (uint64_t)t * pd.in_st;
          if (addr >= n || in[addr]++) return 20 + p;
          if (pd.is_last) {
            const uint64_t K0 = (pd.pidx <= 1) ? ((uint64_t)ct * pd.T + t) : (((uint64_t)ct * pd.T + t) + (uint64_t)pd.n0 * a);
            const uint64_t o = K0 + k * pd.out_sk;
            if (o >= n || out[o]++) return 30 + p;
          } else {
            const uint64_t c = (uint64_t)ct * pd.T + t;
            if (c / pd.cprime >= (1ull << parts[p + 1])) return 40 + p; // jnext is a digit of the next pass
          }
        }
    }
  }
  return 0;
}

extern "C" int plan_check(int max_logn)
{
  for (int logn = 1; logn <= max_logn; logn++) {
    if (int rc = check(logn, 8, 32)) return logn * 100 + rc;  // 31-bit fields
    if (int rc = check(logn, 8, 4)) return logn * 100 + rc;   // 256-bit fields: tiles of at most 4 columns
  }
  return 0;
}

// window groups of the pipelined MSM schedule: a partition of [0, tw) into contiguous non-empty ranges, highest first
extern "C" int msm_groups_check(void)
{
  for (int tw = 1; tw <= 200; tw++)
    for (int want = 1; want <= 20; want++) {
      int lo[MSM_MAX_GROUPS], hi[MSM_MAX_GROUPS];
      const int ng = msm_window_groups(tw, want, lo, hi);
      if (ng < 1 || ng > MSM_MAX_GROUPS || ng > std::max(1, want)) return 1000 * tw + want;
      if (tw < 4 && ng != 1) return -(1000 * tw + want);
      if (hi[0] != tw || lo[ng - 1] != 0) return 2000000 + 1000 * tw + want;
      for (int g = 0; g < ng; g++) {
        if (lo[g] >= hi[g]) return 3000000 + 1000 * tw + want;
        if (g + 1 < ng && lo[g] != hi[g + 1]) return 4000000 + 1000 * tw + want;
      }
    }
  return 0;
}

// shapes of a transform split over P device slots: both factors cover the slots, the factors multiply to N
extern "C" int split_shape_check(void)
{
  for (int logn = 0; logn <= 30; logn++)
    for (int P = 1; P <= 64; P++) {
      SplitShape s;
      const bool ok = split_shape(logn, P, &s);
      const bool pow2 = P >= 2 && (P & (P - 1)) == 0;
      int lp = 0;
      while ((1 << lp) < P)
        lp++;
      if (!pow2 && ok) return 100 * logn + P;
      if (ok && (s.a + s.b != logn || s.a < lp || s.b < lp || s.a < s.b)) return 10000 + 100 * logn + P;
      if (pow2 && !ok && logn >= 2 * lp) return 20000 + 100 * logn + P; // every transform with N >= P^2 can be split
    }
  return 0;
}

// the window plan msm() picks: window size inside the sort's range, never a top window of 1-3 scalar bits, batches of
// small MSMs on the one-level sort, a caller-set c honoured, precompute tables on a c that ignores batch_size
extern "C" int msm_plan_check(void)
{
  for (int bits : {254, 255, 64, 128})
    for (int logn = 0; logn <= 28; logn++)
      for (int batch : {1, 16, 1024})
        for (int pf : {1, 4}) {
          icicle_msm_config_t cfg{};
          cfg.batch_size = batch;
          cfg.precompute_factor = pf;
          const int n = 1 << logn;
          const MsmPlan p = make_plan(n, bits, cfg);
          const int tag = ((bits * 100 + logn) * 10 + (batch > 1)) * 10 + pf;
          if (p.c < 2 || p.c > (p.n_lo > 0 ? 22 : 21)) return tag * 10 + 1;
          if (p.n_lo > 0) { // mixed widths: they add up to the scalar bits, narrow windows below wide ones, full-width scalars only
            if (!p.negate || pf != 1 || batch != 1 || p.bits != bits || p.n_lo >= p.nwin) return tag * 10 + 2;
            if (p.offset(p.nwin - 1) + p.width(p.nwin - 1) != p.bits || p.nb != (1u << (p.c - 1)) || p.wpf != p.nwin) return tag * 10 + 2;
            if (logn < 23) return tag * 10 + 3; // never below 2^23 terms
            continue;
          }
          if (p.negate) return tag * 10 + 2;
          if (p.nwin != (p.bits + 1 + p.c - 1) / p.c || p.wpf != (p.nwin + pf - 1) / pf || p.nb != (1u << (p.c - 1))) return tag * 10 + 2;
          if (pf == 1 && p.nwin > 1 && p.bits > 8 && p.bits + 1 - p.c * (p.nwin - 1) <= 3) return tag * 10 + 3; // tiny top window
          if (pf == 1 && batch > 1 && logn <= 17 && p.c > 11) return tag * 10 + 4;
          // a single small MSM of full-width scalars: the latency rule (8 up to 2^12 terms, then 15 -- or the next c whose top window is no stub)
          if (pf == 1 && batch == 1 && logn < 18 && bits >= 200 && p.c != (logn <= 12 ? 8 : (bits == 255 ? 16 : 15))) return tag * 10 + 8;
          if (pf > 1) { // a base table: msm_precompute_bases and msm must agree whatever batch_size either call carries
            icicle_msm_config_t c1 = cfg;
            c1.batch_size = 1;
            if (make_plan(n, bits, c1).c != p.c) return tag * 10 + 5;
          }
          cfg.c = 13;
          if (make_plan(n, bits, cfg).c != 13) return tag * 10 + 6;
          if (p.seg < 64 || (p.seg & (p.seg - 1)) != 0) return tag * 10 + 7;
        }
  return 0;
}

// ECNTT stage plan + the index algebra of the radix-2^r matrix-form stages (ecntt.hip k_ecntt_terms / k_ecntt_sums), simulated
// over the integers mod a small prime with "points" = residues and "scalar multiplication" = modular product: a transform
// computed stage by stage with ecntt_term_exponent() must equal the O(n^2) definition for every width plan.
static uint64_t mpow(uint64_t b, uint64_t e, uint64_t p)
{
  uint64_t r = 1;
  b %= p;
  while (e) {
    if (e & 1) r = r * b % p;
    b = b * b % p;
    e >>= 1;
  }
  return r;
}
extern "C" int ecntt_plan_check(void)
{
  const uint64_t p = 7681; // 7681 - 1 = 2^9 * 15: roots of unity up to order 512
  const uint64_t g = 17;   // a primitive root mod 7681
  for (int logn = 0; logn <= 9; logn++)
    for (int forced = 0; forced <= 5; forced++)
      for (uint64_t budget : {(uint64_t)16384, (uint64_t)64}) {
        const uint64_t n = (uint64_t)1 << logn;
        int widths[64];
        const int nst = ecntt_stage_plan(logn, n, budget, forced, widths);
        int sum = 0;
        for (int i = 0; i < nst; i++) {
          if (widths[i] < 1 || widths[i] > 5 || (i && widths[i] > widths[i - 1])) return 1000 + logn * 10 + forced;
          if (forced == 0 && widths[i] > 1 && n * ((1ull << widths[i]) - 1) / 2 > budget) return 2000 + logn * 10;
          sum += widths[i];
        }
        if (sum != logn) return 3000 + logn * 10 + forced;
        if (logn == 0) continue;
        const uint64_t w = mpow(g, (p - 1) / n, p); // primitive n-th root
        std::vector<uint64_t> x(n), work(n), next(n);
        for (uint64_t i = 0; i < n; i++)
          x[i] = (i * i * 31 + 7 * i + 3) % p;
        for (uint64_t i = 0; i < n; i++) { // DIT input: bit-reversed
          uint64_t j = 0;
          for (int b = 0; b < logn; b++)
            j |= ((i >> b) & 1) << (logn - 1 - b);
          work[i] = x[j];
        }
        int q0 = 0;
        for (int si = 0; si < nst; si++) {
          const int r = widths[si];
          const uint64_t R = 1ull << r, hr = R >> 1, L = 1ull << q0, M = L * R;
          const uint64_t wM = mpow(w, n / M, p);
          for (uint64_t g0 = 0; g0 < n / R; g0++) {
            const uint64_t pos = g0 & (L - 1), blk = g0 >> q0, base = (blk << (q0 + r)) + pos;
            for (uint64_t u = 0; u < hr; u++) {
              uint64_t ev = work[base], od = 0;
              for (uint64_t j = 1; j < R; j++) {
                const uint64_t t = work[base + j * L] * mpow(wM, ecntt_term_exponent(q0, r, (uint32_t)j, (uint32_t)u, pos), p) % p;
                if (j < hr)
                  ev = (ev + t) % p;
                else
                  od = (od + t) % p;
              }
              next[base + u * L] = (ev + od) % p;
              next[base + (u + hr) * L] = (ev + p - od) % p;
            }
          }
          work.swap(next);
          q0 += r;
        }
        for (uint64_t k = 0; k < n; k++) {
          uint64_t acc = 0;
          for (uint64_t j = 0; j < n; j++)
            acc = (acc + x[j] * mpow(w, j * k % n, p)) % p;
          if (acc != work[k]) return 4000 + logn * 10 + forced;
        }
      }
  return 0;
}
