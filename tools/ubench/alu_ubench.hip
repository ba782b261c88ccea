// Instruction-throughput microbenchmark for gfx950: decides which integer/FP64 building
// block the 256-bit Montgomery multiplier and the 31-bit NTT butterfly should be built on.
// Build: hipcc --offload-arch=gfx950 -O3 alu_ubench.hip -o alu_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8; // independent chains per lane

template <int OP>
__global__ __launch_bounds__(256) void k_alu(uint32_t* out, uint32_t seed)
{
  uint32_t a[CH], b[CH];
  uint64_t c[CH];
  double da[CH], db[CH], dc[CH];
  for (int i = 0; i < CH; i++) {
    a[i] = seed * (threadIdx.x + 1) + i * 7919u;
    b[i] = seed ^ (a[i] * 2654435761u);
    c[i] = ((uint64_t)a[i] << 32) | b[i];
    da[i] = (double)(a[i] & 0xfffff) + 1.0;
    db[i] = (double)(b[i] & 0xfffff) * 1e-9;
    dc[i] = 0.5 + i;
  }
#pragma unroll 1
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      if constexpr (OP == 0) { // v_mad_u64_u32
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
      } else if constexpr (OP == 19) { // v_mad_u64_u32 with an SGPR multiplicand (the m*p half of a Montgomery product)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "s"(seed) : "vcc");
      } else if constexpr (OP == 20) { // dependent chain of v_mad_u64_u32 through ONE accumulator (what mont_asm.hpp issues)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[0]) : "v"(a[i]), "v"(b[i]) : "vcc");
      } else if constexpr (OP == 21) { // v_mov_b32
        asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 22) { // v_lshrrev_b64
        asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(c[i]));
      } else if constexpr (OP == 23) { // v_and_b32
        asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[i]));
      } else if constexpr (OP == 1) { // v_mul_lo_u32
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 2) { // v_mul_hi_u32
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 3) { // v_mul_u32_u24
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 4) { // v_mul_hi_u32_u24
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 5) { // v_mad_u32_u24
        asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 6) { // v_fma_f64
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dc[i]) : "v"(da[i]), "v"(db[i]));
      } else if constexpr (OP == 7) { // v_add_co_u32 + v_addc_co_u32 (64-bit add)
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc"
                     : "+v"(a[i]), "+v"(b[i]) : "v"(b[(i + 1) % CH]), "v"(a[(i + 1) % CH]) : "vcc");
      } else if constexpr (OP == 8) { // v_add_u32
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 9) { // v_mad_u32_u16
        asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 10) { // v_fma_f32
        float fa, fb;
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 11) { // v_dot4_u32_u8
        asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 12) { // v_lshl_add_u64 (64-bit add, 1 instr)
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % CH]));
      } else if constexpr (OP == 13) { // v_mad_i64_i32
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
      } else if constexpr (OP == 14) { // v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dc[i]) : "v"(db[i]));
      } else if constexpr (OP == 15) { // v_cvt_f64_u32 + v_cvt_u32_f64
        asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(dc[i]) : "v"(a[i]));
      } else if constexpr (OP == 16) { // v_add_co_ci only
        asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
      } else if constexpr (OP == 17) { // v_pk_mul_lo_u16
        asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      } else if constexpr (OP == 18) { // v_mul_lo_u16? use v_mad_u16
        asm volatile("v_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < CH; i++) r ^= a[i] ^ b[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32) ^ (uint32_t)dc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, int instr_per_iter, uint32_t* d_out)
{
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_alu<OP><<<blocks, threads>>>(d_out, 12345);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_alu<OP><<<blocks, threads>>>(d_out, 12345);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * threads * ITERS * CH * instr_per_iter; // lane-ops
  double wave_instr = ops / 64.0;
  // cycles per wave-instruction per SIMD: SIMD-cycles available = ms * clk * 1024 SIMDs
  double clk = 2.4e9;
  double cyc = (ms * 1e-3 * clk * 1024.0) / wave_instr;
  printf("%-28s %8.3f ms  %8.2f Glane-ops/s  ~%5.2f cyc/wave-instr/SIMD (at 2.4GHz)\n", name, ms, ops / ms * 1e-6, cyc);
}

// HBM copy bandwidth
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

// random 64B gather bandwidth
__global__ void k_gather(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n_pts, uint32_t mul)
{
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = t; i < n_pts; i += stride) {
    size_t idx = ((i * (size_t)mul) ^ (i >> 7)) % n_pts; // pseudo-random point index
    const uint4* p = in + idx * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.x; acc.z ^= c.x; acc.w += d.x;
  }
  out[t] = acc;
}

// LDS atomic throughput
__global__ __launch_bounds__(1024) void k_lds_atomic(const uint32_t* __restrict__ keys, uint32_t* out, size_t n)
{
  extern __shared__ uint32_t hist[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += stride) acc += atomicAdd(&hist[keys[i] & 32767], 1u);
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hist[threadIdx.x];
}

// global atomic throughput (returning), random addresses over `range` counters
__global__ void k_glb_atomic(const uint32_t* __restrict__ keys, uint32_t* ctr, uint32_t* out, size_t n, uint32_t mask)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += stride) acc += atomicAdd(&ctr[keys[i] & mask], 1u);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d clk=%d MHz  memclk=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000);
  uint32_t* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * 4));
  run<0>("v_mad_u64_u32", 1, d_out);
  run<13>("v_mad_i64_i32", 1, d_out);
  run<19>("v_mad_u64_u32 (sgpr src)", 1, d_out);
  run<20>("v_mad_u64_u32 (1 dep chain)", 1, d_out);
  run<21>("v_mov_b32", 1, d_out);
  run<22>("v_lshrrev_b64", 1, d_out);
  run<23>("v_and_b32 (literal)", 1, d_out);
  run<1>("v_mul_lo_u32", 1, d_out);
  run<2>("v_mul_hi_u32", 1, d_out);
  run<3>("v_mul_u32_u24", 1, d_out);
  run<4>("v_mul_hi_u32_u24", 1, d_out);
  run<5>("v_mad_u32_u24", 1, d_out);
  run<9>("v_mad_u32_u16", 1, d_out);
  run<18>("v_mad_u16", 1, d_out);
  run<17>("v_pk_mul_lo_u16", 1, d_out);
  run<11>("v_dot4_u32_u8", 1, d_out);
  run<6>("v_fma_f64", 1, d_out);
  run<14>("v_mul_f64", 1, d_out);
  run<15>("v_cvt_f64_u32", 1, d_out);
  run<10>("v_fma_f32", 1, d_out);
  run<8>("v_add_u32", 1, d_out);
  run<7>("v_add_co+v_addc (pair)", 2, d_out);
  run<16>("v_addc_co_u32", 1, d_out);
  run<12>("v_lshl_add_u64", 1, d_out);

  // memory tests
  size_t bytes = (size_t)4 << 30;
  uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    k_copy<<<256 * 16, 256>>>(a, b, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("copy 4GiB->4GiB: %.3f ms  %.1f GB/s (R+W)\n", ms, 2.0 * bytes / ms * 1e-6);
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    k_gather<<<256 * 16, 256>>>(a, b, bytes / 64, 2654435761u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("random 64B gather over 4GiB (64M pts): %.3f ms  %.1f GB/s  %.2f Gpts/s\n", ms, (double)bytes / ms * 1e-6, bytes / 64.0 / ms * 1e-6);

  // atomics
  size_t nk = (size_t)1 << 28;
  uint32_t* keys = (uint32_t*)a;
  {
    std::vector<uint32_t> h(1 << 24);
    uint32_t s = 1;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 4; }
    for (size_t off = 0; off < nk; off += h.size()) CK(hipMemcpy(keys + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipFuncSetAttribute((const void*)k_lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    k_lds_atomic<<<256, 1024, 131072>>>(keys, d_out, nk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("LDS returning atomics (32K bins, 128KB): %.3f ms  %.2f Gatomics/s\n", ms, nk / ms * 1e-6);
  uint32_t* ctr = (uint32_t*)b;
  for (uint32_t bits : {15u, 19u, 23u}) {
    CK(hipMemset(ctr, 0, (size_t)4 << bits));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      k_glb_atomic<<<256 * 8, 256>>>(keys, ctr, d_out, nk, (1u << bits) - 1);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("global returning atomics (2^%u counters): %.3f ms  %.2f Gatomics/s\n", bits, ms, nk / ms * 1e-6);
  }
  return 0;
}
