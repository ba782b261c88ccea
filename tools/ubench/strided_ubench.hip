// Strided-tile copy microbenchmark: what HBM bandwidth does the NTT column-pass access pattern allow?
// A block copies a [ROWS x T] tile of u32: ROWS rows spaced `stride` elements apart, T contiguous words per row
// (in place: read then write the same addresses), exactly the traffic of one k_ntt_fast column pass without the
// arithmetic. Build: hipcc --offload-arch=gfx950 -O3 strided_ubench.hip -o strided_ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e = (x);                                                                                                \
    if (e != hipSuccess) {                                                                                             \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);                                                  \
      return 1;                                                                                                        \
    }                                                                                                                  \
  } while (0)

// grid.x = tiles; tile index -> (a, ct): base = a * rows * stride + ct * T ; tiles_per_a = stride / T
template <int RPT> // rows per thread
__global__ __launch_bounds__(512) void k_tile_copy(uint32_t* __restrict__ buf, uint64_t stride, uint32_t T, uint32_t rows, uint32_t tiles_per_a, int write)
{
  const uint32_t a = blockIdx.x / tiles_per_a, ct = blockIdx.x % tiles_per_a;
  uint32_t* base = buf + (uint64_t)a * rows * stride + (uint64_t)ct * T;
  const uint32_t t = threadIdx.x % T, g = threadIdx.x / T; // g in [0, rows / RPT)
  uint32_t v[RPT];
#pragma unroll
  for (int m = 0; m < RPT; m++)
    v[m] = base[(uint64_t)(g + m * (rows / RPT)) * stride + t];
  if (write) {
#pragma unroll
    for (int m = 0; m < RPT; m++)
      base[(uint64_t)(g + m * (rows / RPT)) * stride + t] = v[m] + 1;
  } else {
    uint32_t s = 0;
#pragma unroll
    for (int m = 0; m < RPT; m++)
      s ^= v[m];
    if (s == 0x12345678u) base[t] = s; // never true in practice: keeps the loads alive
  }
}

// the same tile with 16 bytes per lane: a lane owns 4 adjacent columns of a row (T/4 lanes per row)
template <int RPT>
__global__ __launch_bounds__(512) void k_tile_copy_v4(uint32_t* __restrict__ buf, uint64_t stride, uint32_t T, uint32_t rows, uint32_t tiles_per_a, int write)
{
  const uint32_t a = blockIdx.x / tiles_per_a, ct = blockIdx.x % tiles_per_a;
  uint32_t* base = buf + (uint64_t)a * rows * stride + (uint64_t)ct * T;
  const uint32_t T4 = T / 4, t = threadIdx.x % T4, g = threadIdx.x / T4; // g in [0, rows / RPT)
  uint4 v[RPT];
#pragma unroll
  for (int m = 0; m < RPT; m++)
    v[m] = *reinterpret_cast<const uint4*>(base + (uint64_t)(g + m * (rows / RPT)) * stride + 4 * t);
  if (write) {
#pragma unroll
    for (int m = 0; m < RPT; m++) {
      v[m].x += 1;
      *reinterpret_cast<uint4*>(base + (uint64_t)(g + m * (rows / RPT)) * stride + 4 * t) = v[m];
    }
  } else {
    uint32_t s = 0;
#pragma unroll
    for (int m = 0; m < RPT; m++)
      s ^= v[m].x ^ v[m].y ^ v[m].z ^ v[m].w;
    if (s == 0x12345678u) base[t] = s;
  }
}

// the same tile moved the way an LDS-DMA column pass would: global_load_lds_dwordx4 into LDS (no VGPR staging), the next
// batch row's tile in flight while the current one is "processed" (read back from LDS 16 B per lane) and stored
// 16 B per lane. 512 threads, 2 x 32 KiB LDS per block (2 blocks per CU), block = one tile position x `nrows` batch rows.
__global__ __launch_bounds__(512, 4) void k_tile_dma(uint32_t* __restrict__ buf, uint64_t stride, uint32_t tiles_per_a, uint64_t row_elems, uint32_t nrows)
{
  constexpr uint32_t T = 32, ROWS = 256;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][ROWS * T];
  const uint32_t a = blockIdx.x / tiles_per_a, ct = blockIdx.x % tiles_per_a;
  uint32_t* base = buf + (uint64_t)a * ROWS * stride + (uint64_t)ct * T;
  const uint32_t c4 = threadIdx.x % 8, r0 = threadIdx.x / 8; // 8 lanes per 128-B row, 64 rows per sweep
  const uint32_t wave = threadIdx.x / 64;
  auto issue = [&](uint32_t brow, uint32_t which) {
    const uint32_t* src = base + (uint64_t)brow * row_elems;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const uint32_t row = r0 + 64 * m;
      // LDS destination of a wave-instruction: wave-uniform base + lane * 16: rows (8 * wave + 64 * m) .. + 7
      __builtin_amdgcn_global_load_lds(src + (uint64_t)row * stride + 4 * c4, &lds[which][(8 * wave + 64 * m) * T], 16, 0, 0);
    }
  };
  issue(0, 0);
  for (uint32_t b = 0; b < nrows; b++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (b + 1 < nrows) issue(b + 1, (b + 1) & 1);
    uint32_t* dst = base + (uint64_t)b * row_elems;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const uint32_t row = r0 + 64 * m;
      uint4 v = *reinterpret_cast<const uint4*>(&lds[b & 1][row * T + 4 * c4]);
      v.x += 1;
      *reinterpret_cast<uint4*>(dst + (uint64_t)row * stride + 4 * c4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

int main()
{
  const uint64_t n = 1ull << 30; // 4 GiB
  uint32_t* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemset(d, 1, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint32_t rows = 256;
  struct Case {
    uint64_t stride;
    uint32_t T;
    const char* what;
  };
  const Case cases[] = {{65536, 32, "pass 0 (stride 256 KiB, 128 B runs)"}, {256, 32, "pass 1 (stride 1 KiB, 128 B runs)"},
                        {65536, 64, "stride 256 KiB, 256 B runs"},          {256, 64, "stride 1 KiB, 256 B runs"},
                        {32, 32, "contiguous (stride = run)"},
                        {65536, 128, "stride 256 KiB, 512 B runs"},         {256, 128, "stride 1 KiB, 512 B runs"}};
  for (int write = 1; write >= 0; write--)
    for (const Case& c : cases) {
      const uint32_t tiles_per_a = (uint32_t)(c.stride / c.T);
      const uint64_t per_a = (uint64_t)rows * c.stride;
      const uint32_t ntiles = (uint32_t)(n / per_a) * tiles_per_a;
      const unsigned threads = c.T * (rows / 16);
      float best = 1e9;
      for (int it = 0; it < 4; it++) {
        CK(hipEventRecord(e0));
        k_tile_copy<16><<<ntiles, threads>>>(d, c.stride, c.T, rows, tiles_per_a, write);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double bytes = (double)n * 4 * (write ? 2 : 1);
      printf("%-44s %s  4 B/lane  %8.3f ms  %7.0f GB/s\n", c.what, write ? "read+write" : "read only ", best, bytes / best / 1e6);
      // 16 B per lane, same tile; RPT = 16 rows per thread -> threads = (T/4) * rows/16 (>= 64), and RPT = 4 (4x the threads)
      for (int rpt : {16, 4}) {
        const unsigned th4 = (c.T / 4) * (rows / rpt);
        if (th4 < 64 || th4 > 512) continue;
        best = 1e9;
        for (int it = 0; it < 4; it++) {
          CK(hipEventRecord(e0));
          if (rpt == 16) k_tile_copy_v4<16><<<ntiles, th4>>>(d, c.stride, c.T, rows, tiles_per_a, write);
          else k_tile_copy_v4<4><<<ntiles, th4>>>(d, c.stride, c.T, rows, tiles_per_a, write);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("%-44s %s 16 B/lane x%-2d rows/thread %8.3f ms  %7.0f GB/s\n", c.what, write ? "read+write" : "read only ", rpt, best, bytes / best / 1e6);
      }
    }
  { // LDS-DMA variant: 64 batch rows of 2^24 elements, pass-0 and pass-1 patterns, 32 batch rows per block
    const uint64_t row_elems = 1ull << 24;
    const uint32_t batch = (uint32_t)(n / row_elems), rpb = 32;
    for (uint64_t stride : {65536ull, 256ull}) {
      const uint32_t tiles_per_a = (uint32_t)(stride / 32);
      const uint32_t ntiles = (uint32_t)(row_elems / (256 * stride)) * tiles_per_a;
      float best = 1e9;
      for (int it = 0; it < 4; it++) {
        CK(hipEventRecord(e0));
        for (uint32_t g = 0; g < batch / rpb; g++)
          k_tile_dma<<<ntiles, 512>>>(d + (uint64_t)g * rpb * row_elems, stride, tiles_per_a, row_elems, rpb);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("LDS-DMA in, 16 B/lane out, stride %6llu elements, 128 B runs, 32 rows per block   read+write %8.3f ms  %7.0f GB/s\n", (unsigned long long)stride, best, (double)n * 8 / best / 1e6);
    }
  }
  return 0;
}