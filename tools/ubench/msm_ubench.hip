// Round-2 microbenchmarks behind the MSM design decisions. Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icicle_amd/csrc tools/ubench/msm_ubench.hip -o tools/ubench/msm_ubench
//   tools/ubench/msm_ubench [gather|alu|dfma|mfma|all]
// Sections:
//   gather  random 64-byte point gathers/s as a function of the region the indices fall in (64 MiB .. 4 GiB) and of
//           the number of points in flight per lane -- is the 14.9 G/s ceiling a TLB / line / latency limit, and would
//           slicing the base array so that a slice lives in the 256 MiB Infinity Cache lift it?
//   alu     BN254 Fq rates from the PRODUCT code (bigfield.hpp / ec.hpp), operands in registers: modmul, modsqr, the
//           XYZZ mixed add (= the ALU roof of k_accumulate), Fermat inversion, and the per-add work of a batched-affine
//           add (forward product + backward: 5M + 1S), from which the break-even batch size per inversion follows.
//   dfma    the 254-bit integer product as 5x52-bit limbs on v_fma_f64 (hi/lo split, Emmart-style instruction mix)
//           against the 9x29-bit v_mad_u64_u32 column product used by bigfield.hpp.
//   mfma    issue rate of v_mfma_i32_16x16x64_i8 (the "m*p is a constant-matrix product" idea).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "ec.hpp"

using namespace icicle_hip;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static float time_ms(void (*launch)(void*), void* arg, int reps = 3)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch(arg);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0));
    launch(arg);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

// ------------------------------------------------------------------------------------------- gather
__device__ __forceinline__ uint32_t mix(uint32_t x)
{
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t region_mask, uint32_t per_thread, uint32_t seed)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  uint32_t s = mix(t * 2654435761u + seed);
  for (uint32_t i = 0; i < per_thread; i += INFLIGHT) {
    uint4 v[INFLIGHT][4];
#pragma unroll
    for (int q = 0; q < INFLIGHT; q++) {
      s = mix(s + 0x9e3779b9u);
      const uint4* p = in + (size_t)(((uint64_t)s * region_mask) >> 32) * 4; // region_mask = points in the region
      v[q][0] = p[0], v[q][1] = p[1], v[q][2] = p[2], v[q][3] = p[3];
    }
#pragma unroll
    for (int q = 0; q < INFLIGHT; q++) {
      acc.x ^= v[q][0].x ^ v[q][1].y ^ v[q][2].z ^ v[q][3].w;
      acc.y += v[q][0].y + v[q][1].x;
      acc.z ^= v[q][2].x;
      acc.w += v[q][3].x;
    }
  }
  out[t] = acc;
}

struct GatherArgs {
  const uint4* in;
  uint4* out;
  uint32_t mask, per_thread;
  int inflight, blocks;
};
static void launch_gather(void* p)
{
  GatherArgs* a = (GatherArgs*)p;
  if (a->inflight == 1) k_gather<1><<<a->blocks, 256>>>(a->in, a->out, a->mask, a->per_thread, 7);
  if (a->inflight == 2) k_gather<2><<<a->blocks, 256>>>(a->in, a->out, a->mask, a->per_thread, 7);
  if (a->inflight == 4) k_gather<4><<<a->blocks, 256>>>(a->in, a->out, a->mask, a->per_thread, 7);
}

static void bench_gather()
{
  const size_t bytes = (size_t)4 << 30;
  uint4 *a, *o;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&o, (size_t)256 * 32 * 256 * 16));
  CK(hipMemset(a, 1, bytes));
  printf("# random 64-byte gathers (4 x dwordx4 per point), 2^28 gathers per launch\n");
  printf("%-12s %-9s %-8s %10s %12s %10s\n", "region", "inflight", "blocks", "ms", "Ggathers/s", "GB/s");
  for (int mib : {16, 64, 256, 512, 1024, 1536, 2048, 2560, 3072, 4096}) {
    for (int inflight : {1, 4}) {
      for (int bpc : {8}) { // blocks per CU
        GatherArgs g;
        g.in = a;
        g.out = o;
        g.mask = (uint32_t)(((size_t)mib << 20) / 64);
        g.inflight = inflight;
        g.blocks = 256 * bpc;
        const size_t total = (size_t)1 << 28;
        g.per_thread = (uint32_t)(total / ((size_t)g.blocks * 256));
        float ms = time_ms(launch_gather, &g);
        char reg[32];
        snprintf(reg, sizeof reg, "%d MiB", mib);
        printf("%-12s %-9d %-8d %10.3f %12.2f %10.1f\n", reg, inflight, g.blocks, ms, total / ms * 1e-6, total * 64.0 / ms * 1e-6);
      }
    }
  }
  CK(hipFree(a));
  CK(hipFree(o));
}

// ------------------------------------------------------------------------------------------- alu
using E = EC<bn254_g1>;
using F = E::F;
constexpr int ALU_ITERS = 512;

__device__ __forceinline__ F::fe seed_fe(uint32_t s)
{
  F::fe r;
#pragma unroll
  for (int i = 0; i < F::N; i++) {
    s = mix(s + i);
    r.l[i] = s & RB_MASK;
  }
  r.l[F::N - 1] &= 0x1fffff; // < 2^253 < p: an ordinary in-bound element
  return r;
}

template <int OP>
__global__ __launch_bounds__(128, 2) void k_alu(uint32_t* out, uint32_t seed)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  F::fe a = seed_fe(t + seed), b = seed_fe(t * 3 + seed + 1);
  uint32_t r = 0;
  if constexpr (OP == 0) { // modmul chain
#pragma unroll 1
    for (int it = 0; it < ALU_ITERS; it++) {
      a = F::mul(a, b);
      b = F::mul(b, a);
    }
  } else if constexpr (OP == 1) { // modsqr chain
#pragma unroll 1
    for (int it = 0; it < ALU_ITERS; it++) {
      a = F::sqr(a);
      b = F::sqr(b);
    }
  } else if constexpr (OP == 2) { // XYZZ mixed add, operands in registers
    E::XYZZ acc;
    acc.x = a, acc.y = b, acc.zz = seed_fe(t + 5), acc.zzz = seed_fe(t + 9);
    bool empty = false;
    E::Aff p;
    p.x = seed_fe(t + 11), p.y = seed_fe(t + 13);
#pragma unroll 1
    for (int it = 0; it < ALU_ITERS; it++) {
      p.x.l[0] = (p.x.l[0] + 0x1234567u) & RB_MASK; // a different point every step
      p.y.l[1] ^= (uint32_t)it;
      E::madd(acc, empty, p);
    }
    a = acc.x;
    b = F::add(acc.y, F::add(acc.zz, acc.zzz));
    r = empty;
  } else if constexpr (OP == 3) { // Fermat inversion (a^(p-2)), the only inversion the library has
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
      a = F::inv(a);
      a.l[0] ^= 1;
    }
  } else if constexpr (OP == 4) { // batched-affine work per add, WITHOUT the inversion: forward prefix product (1M) +
                                  // backward (inv*prefix 1M, inv*d 1M, lambda 1M, lambda^2 1S, lambda*(x1-x3) 1M)
    F::fe x1 = a, y1 = b, x2 = seed_fe(t + 21), y2 = seed_fe(t + 23), pre = seed_fe(t + 25), inv = seed_fe(t + 27);
#pragma unroll 1
    for (int it = 0; it < ALU_ITERS; it++) {
      F::fe d = F::template sub<2>(x2, x1);
      pre = F::mul(pre, d);                     // forward
      F::fe di = F::mul(inv, pre);              // backward: 1/d_i
      inv = F::mul(inv, d);
      F::fe lam = F::mul(F::template sub<2>(y2, y1), di);
      F::fe x3 = F::template sub<4>(F::sqr(lam), F::add(x1, x2));
      F::fe y3 = F::template sub<2>(F::mul(lam, F::template sub<8>(x1, x3)), y1);
      x1 = x3;                                  // lazily bounded (< 8p), one conditional subtraction each
      F::template cond_sub<4>(x1);
      y1 = y3;
      F::template cond_sub<2>(y1);
      x2.l[0] = (x2.l[0] + 0x765431u) & RB_MASK;
      y2.l[1] ^= (uint32_t)it;

    }
    a = F::add(x1, pre);
    b = F::add(y1, inv);
  }
#pragma unroll
  for (int i = 0; i < F::N; i++)
    r ^= a.l[i] ^ b.l[i];
  out[t] = r;
}

template <int OP>
static void launch_alu(void* p)
{
  k_alu<OP><<<256 * 24, 128>>>((uint32_t*)p, 99);
}

static void bench_alu()
{
  uint32_t* o;
  CK(hipMalloc(&o, (size_t)256 * 24 * 128 * 4));
  const double lanes = 256.0 * 24 * 128;
  float ms;
  ms = time_ms(launch_alu<0>, o);
  const double mul_rate = lanes * ALU_ITERS * 2 / ms * 1e-6;
  printf("BN254 Fq modmul (9x29 limbs, v_mad_u64_u32): %8.3f ms  %8.2f Gmul/s\n", ms, mul_rate);
  ms = time_ms(launch_alu<1>, o);
  printf("BN254 Fq modsqr:                             %8.3f ms  %8.2f Gsqr/s\n", ms, lanes * ALU_ITERS * 2 / ms * 1e-6);
  ms = time_ms(launch_alu<2>, o);
  const double madd_rate = lanes * ALU_ITERS / ms * 1e-6;
  printf("XYZZ mixed add, operands in registers:       %8.3f ms  %8.2f Gadd/s  (= %.2f modmul-equivalents per add)\n", ms, madd_rate, mul_rate / madd_rate);
  ms = time_ms(launch_alu<3>, o);
  const double inv_rate = lanes * 2 / ms * 1e-6;
  printf("Fermat inversion:                            %8.3f ms  %8.3f Ginv/s  (= %.0f modmul-equivalents)\n", ms, inv_rate, mul_rate / inv_rate);
  ms = time_ms(launch_alu<4>, o);
  const double aff_rate = lanes * ALU_ITERS / ms * 1e-6;
  printf("batched-affine add without its inversion:    %8.3f ms  %8.2f Gadd/s  (= %.2f modmul-equivalents per add)\n", ms, aff_rate, mul_rate / aff_rate);
  const double t_madd = 1 / madd_rate, t_aff = 1 / aff_rate, t_inv = 1 / inv_rate;
  printf("break-even adds per (SIMD-replicated) inversion: %.0f   [t_inv / (t_madd - t_affine)]\n", t_inv / (t_madd - t_aff));
  CK(hipFree(o));
}

// ------------------------------------------------------------------------------------------- dfma
// Full 254x254 -> 508-bit product, no reduction. (a) 9 x 29-bit limbs, column sums in one u64 (bigfield.hpp's scheme);
// (b) 5 x 52-bit limbs held in doubles: per limb product  hi = fma(a,b,2^104); lo = fma(a,b,(2^104+2^52)-hi); the two
// bit patterns are accumulated as 64-bit integers (exponent constants removed at the end). Exactness needs RZ mode;
// the instruction mix and cost are the same under the default mode used here.
constexpr int DF_ITERS = 1024;
__global__ __launch_bounds__(256) void k_prod_mad(uint32_t* out, uint32_t seed)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[9], b[9];
  for (int i = 0; i < 9; i++) {
    a[i] = mix(t + seed + i) & RB_MASK;
    b[i] = mix(t * 7 + seed + i) & RB_MASK;
  }
#pragma unroll 1
  for (int it = 0; it < DF_ITERS; it++) {
    uint32_t r[18];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
      const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
      for (int i = lo; i <= hi; i++)
        acc += (uint64_t)a[i] * b[k - i];
      r[k] = (uint32_t)acc & RB_MASK;
      acc >>= RB;
    }
    r[17] = (uint32_t)acc;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      a[i] = r[i] ^ (b[i] >> 1);
      b[i] = r[9 + i] & RB_MASK;
    }
  }
  uint32_t x = 0;
  for (int i = 0; i < 9; i++)
    x ^= a[i] ^ b[i];
  out[t] = x;
}
__global__ __launch_bounds__(256) void k_prod_dfma(uint32_t* out, uint32_t seed)
{
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  double a[5], b[5];
  for (int i = 0; i < 5; i++) {
    a[i] = (double)(((uint64_t)mix(t + seed + i) << 20) | mix(t + i)); // < 2^52
    b[i] = (double)(((uint64_t)mix(t * 7 + seed + i) << 20) | mix(t * 5 + i));
  }
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
#pragma unroll 1
  for (int it = 0; it < DF_ITERS; it++) {
    uint64_t col[11];
#pragma unroll
    for (int k = 0; k < 11; k++)
      col[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const double hi = __builtin_fma(a[i], b[j], C1);
        const double lo = __builtin_fma(a[i], b[j], C2 - hi);
        col[i + j + 1] += (uint64_t)__double_as_longlong(hi);
        col[i + j] += (uint64_t)__double_as_longlong(lo);
      }
    // carry-resolve to 52-bit limbs and convert back to doubles (what a Montgomery step would consume)
    uint64_t carry = 0;
    uint64_t limb[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const uint64_t v = (col[k] & 0x000fffffffffffffull) + carry; // exponent-field constants are dropped by the mask in this proxy
      limb[k] = v & 0x000fffffffffffffull;
      carry = v >> 52;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      a[i] = (double)(long long)(limb[i] ^ (limb[5 + i] >> 1));
      b[i] = (double)(long long)limb[5 + i];
    }
  }
  double s = 0;
  for (int i = 0; i < 5; i++)
    s += a[i] + b[i];
  out[t] = (uint32_t)__double_as_longlong(s);
}
static void launch_mad(void* p) { k_prod_mad<<<256 * 16, 256>>>((uint32_t*)p, 5); }
static void launch_dfma(void* p) { k_prod_dfma<<<256 * 16, 256>>>((uint32_t*)p, 5); }
static void bench_dfma()
{
  uint32_t* o;
  CK(hipMalloc(&o, (size_t)256 * 16 * 256 * 4));
  const double n = 256.0 * 16 * 256 * DF_ITERS;
  float ms = time_ms(launch_mad, o);
  printf("254-bit full product, 9x29 limbs / 81 v_mad_u64_u32:      %8.3f ms  %8.2f Gprod/s\n", ms, n / ms * 1e-6);
  ms = time_ms(launch_dfma, o);
  printf("254-bit full product, 5x52 limbs / 50 v_fma_f64 (hi+lo):  %8.3f ms  %8.2f Gprod/s\n", ms, n / ms * 1e-6);
  CK(hipFree(o));
}

// ------------------------------------------------------------------------------------------- mfma
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int MF_ITERS = 4096;
__global__ __launch_bounds__(256) void k_mfma(int* out)
{
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll 1
  for (int it = 0; it < MF_ITERS; it++) {
    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x + c1.y + c2.z + c3.w;
}
static void launch_mfma(void* p) { k_mfma<<<256 * 8, 256>>>((int*)p); }
static void bench_mfma()
{
  int* o;
  CK(hipMalloc(&o, (size_t)256 * 8 * 256 * 4));
  float ms = time_ms(launch_mfma, o);
  const double instr = 256.0 * 8 * 4 * MF_ITERS * 4; // wave-instructions
  printf("v_mfma_i32_16x16x64_i8: %8.3f ms  %.1f TOPS  %.1f SIMD-cycles per instruction (2.4 GHz)\n", ms, instr * 16 * 16 * 64 * 2 / ms * 1e-9,
         ms * 1e-3 * 2.4e9 * 1024 / instr);
  // m*p for 64 field elements of a wave as M[64 x 37 (7-bit digits)] x Toeplitz(p)[37 x 74]: 4 row tiles x 5 column tiles
  // x 1 K step = 20 MFMA per wave (K = 37 of 64 used) -- plus moving 64 x 37 digits into the A layout and 64 x 74 i32
  // column sums back to their lanes through LDS, and carry-resolving 74 columns per element.
  printf("  -> 20 MFMA per wave-wide m*p = %.0f SIMD-cycles of matrix pipe, against 81 v_mad_u64_u32 = ~494 VALU cycles;\n", 20 * ms * 1e-3 * 2.4e9 * 1024 / instr);
  printf("     the VALU side would still pay digit split (37 ops), LDS transposes (37 + 74 dword moves each way) and 74 carry steps per element.\n");
  CK(hipFree(o));
}

int main(int argc, char** argv)
{
  const char* what = argc > 1 ? argv[1] : "all";
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
  const bool all = !strcmp(what, "all");
  if (all || !strcmp(what, "alu")) bench_alu();
  if (all || !strcmp(what, "dfma")) bench_dfma();
  if (all || !strcmp(what, "mfma")) bench_mfma();
  if (all || !strcmp(what, "gather")) bench_gather();
  return 0;
}
