// Shared device/host helpers for the gfx950 (MI355X) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>

namespace gic {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum DType { DT_F32 = 0, DT_BF16 = 1 };

static constexpr int WAVE = 64;   // CDNA wavefront

// Cache policy (aux) of the trunk's once-read activation loads (LDS-DMA of the A operand in the convolution kernels): 0 = default,
// 2 = nt (streaming: the lines are the first to leave L2).  A build-time switch (GIC_LIB_VARIANT=nt, tools) for measuring whether the
// trunk pass on its stream evicts the decoder's weights from the XCD L2s; the product build leaves it at 0
// unless the measurement says otherwise.
#ifndef GIC_TRUNK_NT
#define GIC_TRUNK_NT 0
#endif

// ---------------------------------------------------------------- status
enum Status {
  GIC_OK = 0,
  GIC_ERR_INVALID_ARG = -1,
  GIC_ERR_UNSUPPORTED = -2,
  GIC_ERR_LAUNCH = -3,
  GIC_ERR_WORKSPACE = -4,
};

void set_last_error(const char* fmt, ...);

#define GIC_CHECK_ARG(cond, ...)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      ::gic::set_last_error(__VA_ARGS__);              \
      return ::gic::GIC_ERR_INVALID_ARG;               \
    }                                                  \
  } while (0)

#define GIC_CHECK_LAUNCH(what)                                                     \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      ::gic::set_last_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return ::gic::GIC_ERR_LAUNCH;                                                \
    }                                                                              \
  } while (0)

#define GIC_PROPAGATE(expr)        \
  do {                             \
    int s__ = (expr);              \
    if (s__ != ::gic::GIC_OK) return s__; \
  } while (0)

// ---------------------------------------------------------------- scalar load/store across dtypes
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float ld_as_f32(const void* p, long i, int dtype) {
  return dtype == DT_F32 ? ((const float*)p)[i] : (float)((const bf16_t*)p)[i];
}
__device__ __forceinline__ void st_from_f32(void* p, long i, int dtype, float v) {
  if (dtype == DT_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = (bf16_t)v;
}
__host__ __device__ __forceinline__ int dtype_size(int dtype) { return dtype == DT_F32 ? 4 : 2; }

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum, result valid in every thread. `red` = LDS scratch of >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---------------------------------------------------------------- counter-based RNG (Philox4x32-10)
// Used for on-device Gumbel uniforms and dropout masks in perf mode; parity
// mode passes explicit noise instead (CPU MT19937 != any device generator).
struct Philox {
  __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  // 4 x 32 random bits for (seed, stream, index)
  __host__ __device__ static inline void gen(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, k0, k1);
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
  // scalar-register form (no arrays: keeps the state out of scratch memory)
  __host__ __device__ static inline void gen4(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t& o0, uint32_t& o1,
                                               uint32_t& o2, uint32_t& o3) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
      const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
      const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
      c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
  }
  // U[0,1) with 24 bits, like torch's float uniform_
  __host__ __device__ static inline float u01(uint32_t bits) { return (bits >> 8) * (1.0f / 16777216.0f); }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-bit key whose unsigned order is (value ascending, index DESCENDING): atomicMax over the keys of a row = its first maximal index
// (the argmax of a caption's Gumbel-perturbed logits across the vocabulary tiles of several workgroups; zero = below every key)
__device__ __forceinline__ unsigned long long row_key(float v, int idx) {
  const unsigned int b = __float_as_uint(v);
  const unsigned int ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)ord << 32) | (unsigned long long)(~(unsigned int)idx);
}
__device__ __forceinline__ int row_key_index(unsigned long long k) { return (int)(~(unsigned int)(k & 0xffffffffull)); }

// Dynamic LDS beyond 64 KB must be granted per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize).  LdsGrant = what the
// kernel already has on each device: one static per kernel instantiation at the call site, so the attribute is set once per device
// (outside any stream capture: every plan runs its first pass eagerly).  The slow path is serialised, so two host threads that need
// different sizes cannot leave the smaller one in force.  false: the runtime refused (the caller falls back or reports).
struct LdsGrant {
  static constexpr int kMaxDev = 16;
  size_t by_dev[kMaxDev];
  std::mutex mu;
  LdsGrant() { for (size_t& g : by_dev) g = 64 * 1024; }
};
template <typename Kf>
static inline bool grant_lds(Kf kernel, size_t bytes, LdsGrant& g) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LdsGrant::kMaxDev) { (void)hipGetLastError(); dev = -1; }
  if (dev >= 0 && bytes <= __atomic_load_n(&g.by_dev[dev], __ATOMIC_ACQUIRE)) return true;
  std::lock_guard<std::mutex> lock(g.mu);
  if (dev >= 0 && bytes <= g.by_dev[dev]) return true;
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (dev >= 0) __atomic_store_n(&g.by_dev[dev], bytes, __ATOMIC_RELEASE);     // an unknown device index: set every time
  return true;
}

// Workgroups go to the 8 XCDs round-robin (blockIdx.x % 8), each XCD with its own 4 MB L2.  xcd_run() renumbers them so that every XCD owns
// ONE contiguous run of the returned index, in its dispatch order: workgroups whose returned indices are neighbours run on the same XCD at
// about the same time, so what they both read (the A rows of one row tile, under its output-channel tiles) comes from HBM / MALL once.
__device__ __forceinline__ int xcd_run(const int b, const int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// host side of the choice "which operand do an XCD's concurrent workgroups share": the output-channel tiles of a row tile as neighbours
// (A once per XCD, every XCD reads all of B) or the row tiles of a channel tile (B once, A once per channel tile).  GIC_XCD_SHARE=a|b forces.
static inline bool xcd_share_a(const long a_bytes, const long b_bytes, const int tiles_n) {
  static const int force = [] { const char* e = getenv("GIC_XCD_SHARE"); return !e ? 0 : (*e == 'a' ? 1 : *e == 'b' ? 2 : 0); }();
  if (tiles_n <= 1) return false;
  if (force) return force == 1;
  return a_bytes * (tiles_n - 1) > 7 * b_bytes;
}

}  // namespace gic
