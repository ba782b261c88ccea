// Element-wise field vector operations next to the NTT path: vector_add / vector_sub / vector_mul,
// scalar_mul_vec and bit_reverse for the NTT fields (BabyBear, KoalaBear, BN254 Fr, BLS12-381 Fr).
// They are what sits between a forward and an inverse NTT in a polynomial product (NTT -> point-wise
// product -> inverse NTT) or around the kNR / kRN orderings, so that such a pipeline stays on the device.
//
// Reference semantics: icicle/src/vec_ops.cpp:71-84 (vector_add), :136-149 (vector_sub), :169-182
// (vector_mul), :362-366 (scalar_mul_vec; scalar_add_vec / scalar_sub_vec = scalar op vector next to it), :440-444
// (bit_reverse); CPU backend
// icicle/backend/cpu/src/field/cpu_vec_ops.cpp:325-342 (one scalar per batch entry, stride = batch for
// columns_batch), :536-560 (bit reverse per batch entry). Values are plain (non-Montgomery) residues and
// stay so: products are computed as montmul(montmul(a, b), R^2).
// All of these are HBM-bound streams (2 reads + 1 write per element, or 1 + 1 for bit_reverse).
#include "common.h"
#include "bigfield.hpp"
#include "smallfield.hpp"
#include "goldfield.hpp"

namespace icicle_hip {

  enum { VOP_ADD = 0, VOP_SUB = 1, VOP_MUL = 2 };

  template <class PR>
  struct SmallElem {
    using S = SmallField<PR>;
    static constexpr int W = 1;
    struct T {
      uint32_t v;
    };
    static __device__ __forceinline__ T load(const uint32_t* p) { return T{*p}; }
    static __device__ __forceinline__ void store(uint32_t* p, const T& x) { *p = x.v; }
    static __device__ __forceinline__ T apply(int op, const T& a, const T& b)
    {
      if (op == VOP_ADD) return T{S::add(a.v, b.v)};
      if (op == VOP_SUB) return T{S::sub(a.v, b.v)};
      return T{S::mul(S::mul(a.v, b.v), PR::R2)};
    }
  };

  template <class PR>
  struct BigElem {
    using F = FieldOps<PR>;
    static constexpr int W = F::N32;
    using T = typename F::fe;
    static __device__ __forceinline__ T load(const uint32_t* p)
    {
      uint32_t w[W];
#pragma unroll
      for (int i = 0; i < W; i++)
        w[i] = p[i];
      return F::unpack(w);
    }
    static __device__ __forceinline__ void store(uint32_t* p, const T& x)
    {
      uint32_t w[W];
      F::pack(w, F::reduce(x));
#pragma unroll
      for (int i = 0; i < W; i++)
        p[i] = w[i];
    }
    static __device__ __forceinline__ T apply(int op, const T& a, const T& b)
    {
      if (op == VOP_ADD) return F::add(a, b);
      if (op == VOP_SUB) return F::template sub<2>(a, b);
      return F::mul(F::mul(a, b), F::from_const(PR::R2));
    }
  };

  struct GoldElem { // goldilocks: canonical 64-bit values, goldfield.hpp
    using F = FieldOps<goldilocks_params>;
    static constexpr int W = 2;
    using T = GoldFe;
    static __device__ __forceinline__ T load(const uint32_t* p) { return F::unpack(p); }
    static __device__ __forceinline__ void store(uint32_t* p, const T& x) { F::pack(p, x); }
    static __device__ __forceinline__ T apply(int op, const T& a, const T& b)
    {
      if (op == VOP_ADD) return F::add(a, b);
      if (op == VOP_SUB) return F::template sub<2>(a, b);
      return F::mul(a, b);
    }
  };

  // out[i] = a[ai(i)] op b[i]; a_scalar: a holds one scalar per batch entry
  template <class EL>
  __global__ __launch_bounds__(256) void k_vec2(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, uint64_t total, int op, bool a_scalar, uint64_t size, uint32_t batch, bool columns)
  {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
      const uint64_t ai = !a_scalar ? i : (columns ? i % batch : i / size);
      EL::store(out + i * EL::W, EL::apply(op, EL::load(a + ai * EL::W), EL::load(b + i * EL::W)));
    }
  }

  // out[b][bitrev(j)] = in[b][j]; element j of entry b at (columns ? j*batch + b : b*size + j)
  template <int W>
  __global__ __launch_bounds__(256) void k_bitrev_elems(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t logn, uint32_t batch, bool columns)
  {
    const uint64_t n = (uint64_t)1 << logn;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n * batch) return;
    const uint64_t j = columns ? t / batch : t % n, b = columns ? t % batch : t / n;
    const uint64_t r = logn == 0 ? 0 : (__brevll(j) >> (64 - logn));
    const uint64_t src = columns ? j * batch + b : b * n + j, dst = columns ? r * batch + b : b * n + r;
    uint32_t w[W];
#pragma unroll
    for (int i = 0; i < W; i++)
      w[i] = in[src * W + i];
#pragma unroll
    for (int i = 0; i < W; i++)
      out[dst * W + i] = w[i];
  }

  struct Staged { // host <-> device staging of one operand
    TempBuf buf;
    const uint32_t* dev = nullptr;
    icicle_error_t in(const void* p, bool on_device, size_t bytes, hipStream_t st)
    {
      if (on_device) {
        dev = (const uint32_t*)p;
        return ICICLE_SUCCESS;
      }
      HIP_TRY(buf.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      HIP_TRY(hipMemcpyAsync(buf.ptr(), p, bytes, hipMemcpyHostToDevice, st), ICICLE_COPY_FAILED);
      dev = buf.as<uint32_t>();
      return ICICLE_SUCCESS;
    }
  };

  static icicle_error_t finish_out(void* output, uint32_t* d_out, bool on_device, bool is_async, size_t bytes, hipStream_t st)
  {
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(output, d_out, bytes, hipMemcpyDeviceToHost, st), ICICLE_COPY_FAILED);
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    } else if (!is_async) {
      HIP_TRY(hipStreamSynchronize(st), ICICLE_SYNCHRONIZATION_FAILED);
    }
    return ICICLE_SUCCESS;
  }

  template <class EL>
  static icicle_error_t vec2_run(const void* a, const void* b, uint64_t size, const icicle_vec_ops_config_t* cfg, void* out, int op, bool a_scalar)
  {
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (size == 0) return ICICLE_SUCCESS;
    if (!a || !b || !out) return ICICLE_INVALID_POINTER;
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const uint32_t batch = (uint32_t)std::max(1, cfg->batch_size);
    const uint64_t total = size * batch;
    const size_t ebytes = (size_t)EL::W * 4, bytes = (size_t)total * ebytes;
    Staged sa, sb;
    ICICLE_TRY(sa.in(a, cfg->is_a_on_device, a_scalar ? (size_t)batch * ebytes : bytes, st));
    ICICLE_TRY(sb.in(b, cfg->is_b_on_device, bytes, st));
    TempBuf d_out_tmp;
    uint32_t* d_out = (uint32_t*)out;
    if (!cfg->is_result_on_device) {
      HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      d_out = d_out_tmp.as<uint32_t>();
    }
    const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, 1u << 20);
    k_vec2<EL><<<grid, 256, 0, st>>>(sa.dev, sb.dev, d_out, total, op, a_scalar, size, batch, cfg->columns_batch);
    LAUNCH_CHECK("k_vec2", st);
    return finish_out(out, d_out, cfg->is_result_on_device, cfg->is_async, bytes, st);
  }

  template <int W>
  static icicle_error_t bitrev_run(const void* in, uint64_t size, const icicle_vec_ops_config_t* cfg, void* out)
  {
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (!in || !out) return ICICLE_INVALID_POINTER;
    if (size == 0 || (size & (size - 1)) != 0) return ICICLE_INVALID_ARGUMENT; // cpu_vec_ops.cpp:540-543
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const uint32_t batch = (uint32_t)std::max(1, cfg->batch_size);
    uint32_t logn = 0;
    while (((uint64_t)1 << logn) < size)
      logn++;
    const size_t bytes = (size_t)size * batch * W * 4;
    Staged si;
    ICICLE_TRY(si.in(in, cfg->is_a_on_device, bytes, st));
    TempBuf d_out_tmp;
    uint32_t* d_out = (uint32_t*)out;
    // out of place on the device; an in-place call (same device pointer) goes through a temporary
    const bool need_tmp = !cfg->is_result_on_device || (const void*)si.dev == (const void*)out;
    if (need_tmp) {
      HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      d_out = d_out_tmp.as<uint32_t>();
    }
    const uint64_t tot = size * batch;
    k_bitrev_elems<W><<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(si.dev, d_out, logn, batch, cfg->columns_batch);
    LAUNCH_CHECK("k_bitrev_elems", st);
    if (cfg->is_result_on_device && need_tmp) {
      HIP_TRY(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
      d_out = (uint32_t*)out;
    }
    return finish_out(out, d_out, cfg->is_result_on_device, cfg->is_async, bytes, st);
  }

  // ---- matrix_transpose (icicle/src/matrix_ops.cpp:73-102, backend/cpu/src/field/cpu_matrix_ops.cpp:175-201,333-362) ----------
  // out[b][c][r] = in[b][r][c] for batch_size row-major nof_rows x nof_cols matrices of W-word elements. 32 x 32-element
  // tiles through LDS: both sides move runs of 32 * W contiguous words. The Rust NTT suite calls it on the main device
  // around every columns_batch transform (wrappers/rust/icicle-core/src/ntt/tests.rs:311-335).
  template <int W>
  __global__ __launch_bounds__(256) void k_transpose(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t rows, uint32_t cols, uint32_t tiles_c, uint32_t batch)
  {
    constexpr uint32_t RW = 32 * W; // words per tile row
    __shared__ uint32_t tile[32][RW + 1];
    const uint32_t tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const uint32_t r0 = tr * 32, c0 = tc * 32;
    const uint32_t nr = min(32u, rows - r0), nc = min(32u, cols - c0);
    const uint64_t mat_words = (uint64_t)rows * cols * W;
    for (uint32_t b = blockIdx.y; b < batch; b += gridDim.y) {
      const uint32_t* pi = in + b * mat_words + ((uint64_t)r0 * cols + c0) * W;
      uint32_t* po = out + b * mat_words + ((uint64_t)c0 * rows + r0) * W;
#pragma unroll
      for (uint32_t e = threadIdx.x; e < 32 * RW; e += 256) {
        const uint32_t r = e / RW, k = e % RW;
        if (r < nr && k < nc * W) tile[r][k] = pi[(uint64_t)r * cols * W + k];
      }
      __syncthreads();
#pragma unroll
      for (uint32_t e = threadIdx.x; e < 32 * RW; e += 256) {
        const uint32_t c = e / RW, k = e % RW, r = k / W, w = k % W;
        if (c < nc && r < nr) po[(uint64_t)c * rows * W + k] = tile[r][c * W + w];
      }
      __syncthreads();
    }
  }

  template <int W>
  static icicle_error_t transpose_run(const void* in, uint32_t rows, uint32_t cols, const icicle_vec_ops_config_t* cfg, void* out)
  {
    if (!cfg) return ICICLE_INVALID_POINTER;
    if (!in || !out || rows == 0 || cols == 0) return ICICLE_INVALID_ARGUMENT; // cpu_matrix_ops.cpp:337-340
    if (cfg->columns_batch) return ICICLE_INVALID_ARGUMENT;                    // :342-345
    ICICLE_TRY(bind_current_device());
    hipStream_t st = (hipStream_t)cfg->stream;
    const uint32_t batch = (uint32_t)std::max(1, cfg->batch_size);
    const size_t bytes = (size_t)rows * cols * batch * W * 4;
    Staged si;
    ICICLE_TRY(si.in(in, cfg->is_a_on_device, bytes, st));
    TempBuf d_out_tmp;
    uint32_t* d_out = (uint32_t*)out;
    // out of place on the device; an in-place call (cpu_matrix_ops.cpp:348-359 supports it) goes through a temporary
    const bool need_tmp = !cfg->is_result_on_device || (const void*)si.dev == (const void*)out;
    if (need_tmp) {
      HIP_TRY(d_out_tmp.alloc(bytes, st), ICICLE_ALLOCATION_FAILED);
      d_out = d_out_tmp.as<uint32_t>();
    }
    const uint64_t tiles_r = ((uint64_t)rows + 31) / 32, tiles_c = ((uint64_t)cols + 31) / 32;
    if (tiles_r * tiles_c > 0x7fffffffull) return ICICLE_INVALID_ARGUMENT;
    k_transpose<W><<<dim3((unsigned)(tiles_r * tiles_c), std::min(batch, 65535u)), 256, 0, st>>>(si.dev, d_out, rows, cols, (uint32_t)tiles_c, batch);
    LAUNCH_CHECK("k_transpose", st);
    if (cfg->is_result_on_device && need_tmp) {
      HIP_TRY(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToDevice, st), ICICLE_COPY_FAILED);
      d_out = (uint32_t*)out;
    }
    return finish_out(out, d_out, cfg->is_result_on_device, cfg->is_async, bytes, st);
  }

} // namespace icicle_hip

using namespace icicle_hip;

#define GUARDED(expr)                                                                                                  \
  try {                                                                                                                \
    return (expr);                                                                                                     \
  } catch (...) {                                                                                                      \
    return ICICLE_INVALID_ARGUMENT;                                                                                    \
  }

#define DEFINE_VEC_ARITH(NAME, EL, W)                                                                                  \
  extern "C" icicle_error_t NAME##_vector_add(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_ADD, false))); } \
  extern "C" icicle_error_t NAME##_vector_sub(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_SUB, false))); } \
  extern "C" icicle_error_t NAME##_vector_mul(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_MUL, false))); } \
  extern "C" icicle_error_t NAME##_scalar_mul_vec(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_MUL, true))); } \
  extern "C" icicle_error_t NAME##_scalar_add_vec(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_ADD, true))); } \
  extern "C" icicle_error_t NAME##_scalar_sub_vec(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_SUB, true))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_scalar_add_vec(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_ADD, true))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_scalar_sub_vec(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_SUB, true))); } \
  extern "C" icicle_error_t NAME##_bit_reverse(const void* i, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((bitrev_run<W>(i, n, c, o))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_vector_add(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_ADD, false))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_vector_sub(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_SUB, false))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_vector_mul(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_MUL, false))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_scalar_mul_vec(const void* a, const void* b, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((vec2_run<EL>(a, b, n, c, o, VOP_MUL, true))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_bit_reverse(const void* i, uint64_t n, const icicle_vec_ops_config_t* c, void* o) { GUARDED((bitrev_run<W>(i, n, c, o))); }

DEFINE_VEC_ARITH(babybear, SmallElem<babybear_params>, 1)
DEFINE_VEC_ARITH(koalabear, SmallElem<koalabear_params>, 1)
DEFINE_VEC_ARITH(bn254, BigElem<bn254_fr_params>, 8)
DEFINE_VEC_ARITH(bls12_381, BigElem<bls12_381_fr_params>, 8)
DEFINE_VEC_ARITH(bls12_377, BigElem<bls12_377_fr_params>, 8)
DEFINE_VEC_ARITH(grumpkin, BigElem<bn254_fq_params>, 8)
DEFINE_VEC_ARITH(stark252, BigElem<stark252_fr_params>, 8)
DEFINE_VEC_ARITH(goldilocks, GoldElem, 2)

// matrix_transpose / extension_matrix_transpose (icicle/src/matrix_ops.cpp:75-102): W = u32 words per element
#define DEFINE_TRANSPOSE(NAME, SUFFIX, W)                                                                              \
  extern "C" icicle_error_t NAME##_##SUFFIX(const void* i, uint32_t r, uint32_t c, const icicle_vec_ops_config_t* cfg, void* o) { GUARDED((transpose_run<W>(i, r, c, cfg, o))); } \
  extern "C" icicle_error_t icicle_hip_##NAME##_##SUFFIX(const void* i, uint32_t r, uint32_t c, const icicle_vec_ops_config_t* cfg, void* o) { GUARDED((transpose_run<W>(i, r, c, cfg, o))); }
DEFINE_TRANSPOSE(babybear, matrix_transpose, 1)
DEFINE_TRANSPOSE(koalabear, matrix_transpose, 1)
DEFINE_TRANSPOSE(babybear, extension_matrix_transpose, 4)
DEFINE_TRANSPOSE(koalabear, extension_matrix_transpose, 4)
DEFINE_TRANSPOSE(goldilocks, matrix_transpose, 2)
DEFINE_TRANSPOSE(goldilocks, extension_matrix_transpose, 4)
DEFINE_TRANSPOSE(bn254, matrix_transpose, 8)
DEFINE_TRANSPOSE(bls12_381, matrix_transpose, 8)
DEFINE_TRANSPOSE(bls12_377, matrix_transpose, 8)
DEFINE_TRANSPOSE(grumpkin, matrix_transpose, 8)
DEFINE_TRANSPOSE(stark252, matrix_transpose, 8)
