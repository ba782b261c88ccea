// LDS atomic / staging microbenchmark: what bounds the counting-sort passes of the MSM?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE> // 0: returning atomic, 1: non-returning atomic, 2: plain ds_write to computed slot, 3: read keys only
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ keys, uint32_t* out, size_t n, uint32_t mask)
{
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n; i += stride) {
    const uint32_t kk = keys[i] & mask;
    if (MODE == 0) acc += atomicAdd(&lds[kk], 1u);
    else if (MODE == 1) atomicAdd(&lds[kk], 1u);
    else if (MODE == 2) lds[kk] = (uint32_t)i;
    else acc += kk;
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
void run(const char* name, const uint32_t* keys, uint32_t* out, size_t n, uint32_t bins)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    k<MODE><<<1024, 1024, 65536>>>(keys, out, n, bins - 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("%-28s bins=%5u  %.3f ms  %.1f G elem/s\n", name, bins, ms, n / ms * 1e-6);
}

int main()
{
  const size_t n = (size_t)1 << 29;
  uint32_t *keys, *out;
  CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&out, 1024 * 1024 * 4));
  {
    std::vector<uint32_t> h(1 << 24);
    uint32_t s = 1;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 7; }
    for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(keys + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  for (uint32_t bins : {128u, 512u, 1024u, 16384u}) {
    run<3>("read keys only", keys, out, n, bins);
    run<0>("returning ds_add", keys, out, n, bins);
    run<1>("non-returning ds_add", keys, out, n, bins);
    run<2>("plain ds_write", keys, out, n, bins);
  }
  return 0;
}
