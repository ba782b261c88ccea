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


// ---- two-pass question: a 2^24 transform as 2^12 x 2^12 needs [4096 rows x T columns] tiles held
// on chip; 1024 threads x 64 words = 4096 x 16, so T = 16 and every HBM run is 64 bytes (half a line). This is that
// tile moved with 16-byte lanes (4 lanes per run, 16 runs per thread), in place (pass-0 pattern: rows 16 KiB apart), or
// read from one contiguous 256 KiB chunk and written strided (last-pass pattern). SWZ: blocks b and b + 8 -- same XCD,
// dispatched back to back -- take the two halves of the same 128-byte lines, so the second half can hit in that L2.
template <bool SWZ, bool CONTIG_READ>
__global__ __launch_bounds__(1024) void k_tile_2pass(uint32_t* __restrict__ buf, int write)
{
  constexpr uint32_t ROWS = 4096, T = 16, TPA = 4096 / T; // tiles per 2^24-element transform
  uint32_t tile = blockIdx.x;
  if (SWZ) {
    const uint32_t xcd = blockIdx.x % 8, k = blockIdx.x / 8;
    tile = ((k / 2) * 8 + xcd) * 2 + (k % 2);
  }
  const uint32_t a = tile / TPA, ct = tile % TPA;
  uint32_t* tbase = buf + ((uint64_t)a << 24);
  const uint32_t t = threadIdx.x % 4, g = threadIdx.x / 4; // 256 row groups
  uint4 v[16];
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const uint32_t row = g + 256 * m;
    const uint32_t* src = CONTIG_READ ? tbase + (uint64_t)ct * (ROWS * T) + row * T + 4 * t : tbase + (uint64_t)row * 4096 + ct * T + 4 * t;
    v[m] = *reinterpret_cast<const uint4*>(src);
  }
  if (write) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
      const uint32_t row = g + 256 * m;
      v[m].x += 1;
      *reinterpret_cast<uint4*>(tbase + (uint64_t)row * 4096 + ct * T + 4 * t) = v[m];
    }
  } else {
    uint32_t s = 0;
#pragma unroll
    for (int m = 0; m < 16; m++)
      s ^= v[m].x ^ v[m].y ^ v[m].z ^ v[m].w;
    if (s == 0x12345678u) tbase[t] = s;
  }
}

// in-place contiguous read + write over a region that is swept `reps` times inside ONE launch sequence: does a working
// set that fits the 256 MiB Infinity Cache move faster than HBM? (row-group question: 2 rows of 2^24 words = 128 MiB)
__global__ __launch_bounds__(256) void k_sweep(uint4* __restrict__ buf, uint64_t n4)
{
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  uint4 v = buf[i];
  v.x += 1;
  buf[i] = v;
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
  { // 64-byte runs: the [4096 x 16] register tile of a two-pass plan
    const uint32_t ntiles = (uint32_t)(n >> 24) * 256;
    for (int write = 1; write >= 0; write--)
      for (int variant = 0; variant < 4; variant++) {
        float best = 1e9;
        for (int it = 0; it < 4; it++) {
          CK(hipEventRecord(e0));
          switch (variant) {
          case 0: k_tile_2pass<false, false><<<ntiles, 1024>>>(d, write); break;
          case 1: k_tile_2pass<true, false><<<ntiles, 1024>>>(d, write); break;
          case 2: k_tile_2pass<false, true><<<ntiles, 1024>>>(d, write); break;
          default: k_tile_2pass<true, true><<<ntiles, 1024>>>(d, write); break;
          }
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        const double bytes = (double)n * 4 * (write ? 2 : 1);
        printf("two-pass tile 4096 x 16 (64 B runs, 16 B/lane, 1024 threads) %-28s %-10s %s %8.3f ms  %7.0f GB/s\n",
               variant >= 2 ? "contiguous read, strided write" : "strided in place (16 KiB rows)", (variant & 1) ? "XCD-paired" : "linear", write ? "read+write" : "read only ", best,
               bytes / best / 1e6);
      }
  }
  { // Infinity-Cache question: the same bytes moved as 16 sweeps over a small region vs one sweep over 4 GiB
    for (uint64_t mb : {32ull, 64ull, 128ull, 192ull, 256ull, 512ull, 4096ull}) {
      const uint64_t n4 = (mb << 20) / 16, reps = 4096 / mb * 4;
      float best = 1e9;
      for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0));
        for (uint64_t r = 0; r < reps; r++)
          k_sweep<<<(unsigned)((n4 + 255) / 256), 256>>>(reinterpret_cast<uint4*>(d), n4);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("in-place sweep of a %4llu MiB region x %3llu launches: %8.3f ms  %7.0f GB/s read+write\n", (unsigned long long)mb, (unsigned long long)reps, best,
             (double)(mb << 20) * 2 * reps / best / 1e6);
    }
  }
  return 0;
}
