// Small memory-bound helpers + the generic C-ABI entry points (error string, gemm, cast, embedding).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gicap.h"
#include "kernels.h"

namespace gic {

static thread_local char g_last_error[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

namespace {

// ---- 2-D cast/copy: one thread per element, coalesced along columns
// out[c][r] = in[r][c] for a row-major [rows, cols] matrix, 32x32 tiles through LDS (both sides coalesced)
template <typename TA>
__global__ void transpose2d_kernel(const TA* __restrict__ in, TA* __restrict__ out, int rows, int cols) {
  __shared__ TA tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[(long)(r0 + j) * cols + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < cols && r0 + tx < rows) out[(long)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

template <typename TS, typename TD>
__global__ void cast2d_kernel(const TS* __restrict__ src, long lds, TD* __restrict__ dst, long ldd, long rows, long cols) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i % cols;
    dst[r * ldd + c] = from_f32<TD>(to_f32<TS>(src[r * lds + c]));
  }
}

// f32 -> bf16 for 16-byte aligned operands with cols % 8 == 0: a thread converts 8 adjacent columns (two 16-byte loads, one 16-byte store),
// two such pieces per iteration in flight (the weight images of the compute dtype are refreshed every step: 60 MB of them)
__global__ __launch_bounds__(256) void cast2d_f32_bf16_vec_kernel(const float* __restrict__ src, long lds, bf16_t* __restrict__ dst, long ldd,
                                                                  long rows, long cols) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const long cpr = cols / 8, total = rows * cpr;
  const long S = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += 2 * S) {
    f4 a[2][2];
    long o[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long iu = i + u * S < total ? i + u * S : i;
      const long r = iu / cpr, c = (iu - r * cpr) * 8;
      o[u] = r * ldd + c;
      const float* p = src + r * lds + c;
      a[u][0] = *(const f4*)p;
      a[u][1] = *(const f4*)(p + 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (i + u * S >= total) break;
      bf16x8 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = (bf16_t)a[u][0][k]; v[4 + k] = (bf16_t)a[u][1][k]; }
      *(bf16x8*)(dst + o[u]) = v;
    }
  }
}

// ---- column sums: block = 64 columns x 4 row-lanes; rows strided over gridDim.y blocks, f32 atomics across them
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ A, long lda, long rows, long cols,
                                                       float* __restrict__ out, float* __restrict__ out2) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < cols)
    for (long r = (long)blockIdx.y * 4 + ry; r < rows; r += (long)gridDim.y * 4) s += to_f32<T>(A[r * lda + c]);
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < cols) {
    const float t = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
    atomicAdd(&out[c], t);
    if (out2) atomicAdd(&out2[c], t);
  }
}

// ---- the same for 16-byte aligned operands: a thread sums 8 (bf16) / 4 (f32) adjacent columns, 16 bytes per load, four rows in flight
// per thread; block = 32 column groups x 8 row lanes.  (The scalar kernel above keeps one 2-byte load per thread in flight.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ A, long lda, long rows, long cols,
                                                           float* __restrict__ out, float* __restrict__ out2) {
  constexpr int VN = 16 / (int)sizeof(T);
  __shared__ float red[8][32][VN + 1];
  const int cg = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const long c0 = ((long)blockIdx.x * 32 + cg) * VN;
  float s[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) s[k] = 0.f;
  if (c0 < cols) {
    const long step = (long)gridDim.y * 8;
    for (long r = (long)blockIdx.y * 8 + ry; r < rows; r += 4 * step) {
      T v[4][VN];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long ru = r + u * step;
        *(uint4*)v[u] = *(const uint4*)(A + (ru < rows ? ru : r) * lda + c0);      // rows past the end re-read row r (weight 0 below)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float wgt = r + u * step < rows ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < VN; ++k) s[k] += wgt * to_f32<T>(v[u][k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < VN; ++k) red[ry][cg][k] = s[k];
  __syncthreads();
  if (ry == 0 && c0 < cols) {
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q][cg][k];
      atomicAdd(&out[c0 + k], t);
      if (out2) atomicAdd(&out2[c0 + k], t);
    }
  }
}

__global__ void embedding_fwd_kernel(const float* __restrict__ w, const int64_t* __restrict__ ids, float* __restrict__ out,
                                     long n, int V, int E) {
  const long total = n * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / E;
    const int e = (int)(i % E);
    long id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[i] = w[id * E + e];
  }
}

__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids,
                                     float* __restrict__ dw, long n, int V, int E) {
  const long total = n * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / E;
    const int e = (int)(i % E);
    long id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    atomicAdd(&dw[id * E + e], dout[i]);
  }
}

inline int grid_for(long total, int block = 256, int cap = 2048) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

int transpose2d(const void* src, void* dst, int dtype, long rows, long cols, hipStream_t stream) {
  GIC_CHECK_ARG(src && dst && rows > 0 && cols > 0, "transpose2d: bad argument");
  const dim3 tg((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32));
  if (dtype == DT_F32)
    hipLaunchKernelGGL((transpose2d_kernel<float>), tg, dim3(256), 0, stream, (const float*)src, (float*)dst, (int)rows, (int)cols);
  else
    hipLaunchKernelGGL((transpose2d_kernel<bf16_t>), tg, dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, (int)rows, (int)cols);
  GIC_CHECK_LAUNCH("transpose2d");
  return GIC_OK;
}

int fill_zero(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return GIC_OK;
  hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
  if (e != hipSuccess) {
    set_last_error("memset failed: %s", hipGetErrorString(e));
    return GIC_ERR_LAUNCH;
  }
  return GIC_OK;
}

int cast2d(const void* src, int sdt, long lds, void* dst, int ddt, long ldd, long rows, long cols, hipStream_t stream) {
  GIC_CHECK_ARG(src && dst, "cast2d: null pointer");
  if (rows * cols == 0) return GIC_OK;
  const int g = grid_for(rows * cols);
  if (sdt == DT_F32 && ddt == DT_F32)
    hipLaunchKernelGGL((cast2d_kernel<float, float>), dim3(g), dim3(256), 0, stream, (const float*)src, lds, (float*)dst, ldd, rows, cols);
  else if (sdt == DT_F32 && ddt == DT_BF16 && cols % 8 == 0 && lds % 4 == 0 && ldd % 8 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0)
    hipLaunchKernelGGL(cast2d_f32_bf16_vec_kernel, dim3(grid_for(rows * cols / 16)), dim3(256), 0, stream, (const float*)src, lds, (bf16_t*)dst, ldd,
                       rows, cols);
  else if (sdt == DT_F32 && ddt == DT_BF16)
    hipLaunchKernelGGL((cast2d_kernel<float, bf16_t>), dim3(g), dim3(256), 0, stream, (const float*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else if (sdt == DT_BF16 && ddt == DT_F32)
    hipLaunchKernelGGL((cast2d_kernel<bf16_t, float>), dim3(g), dim3(256), 0, stream, (const bf16_t*)src, lds, (float*)dst, ldd, rows, cols);
  else if (sdt == DT_BF16 && ddt == DT_BF16)
    hipLaunchKernelGGL((cast2d_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, stream, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else {
    set_last_error("cast2d: bad dtypes %d -> %d", sdt, ddt);
    return GIC_ERR_UNSUPPORTED;
  }
  GIC_CHECK_LAUNCH("cast2d");
  return GIC_OK;
}

int colsum(const void* A, int dtype, long lda, long rows, long cols, float* out, float* out2, int accumulate,
           hipStream_t stream) {
  GIC_CHECK_ARG(A && out, "colsum: null pointer");
  if (cols == 0) return GIC_OK;
  if (!accumulate) {
    GIC_PROPAGATE(fill_zero(out, cols * sizeof(float), stream));
    if (out2) GIC_PROPAGATE(fill_zero(out2, cols * sizeof(float), stream));
  }
  if (rows == 0) return GIC_OK;
  const int vn = 16 / dtype_size(dtype);
  if (cols % vn == 0 && lda % vn == 0 && (((uintptr_t)A) & 15) == 0) {
    const int gx = cdiv(cols / vn, 32);
    int gy = cdiv(rows, 8 * 8);
    const int want = 2048 / (gx < 1 ? 1 : gx);
    gy = gy > want ? (want < 1 ? 1 : want) : gy;
    if (dtype == DT_F32)
      hipLaunchKernelGGL((colsum_vec_kernel<float>), dim3(gx, gy), dim3(256), 0, stream, (const float*)A, lda, rows, cols, out, out2);
    else
      hipLaunchKernelGGL((colsum_vec_kernel<bf16_t>), dim3(gx, gy), dim3(256), 0, stream, (const bf16_t*)A, lda, rows, cols, out, out2);
    GIC_CHECK_LAUNCH("colsum");
    return GIC_OK;
  }
  const int gx = cdiv(cols, 64);
  int gy = cdiv(rows, 4 * 16);
  const int want = 1024 / (gx < 1 ? 1 : gx);
  gy = gy > want ? (want < 1 ? 1 : want) : gy;
  if (dtype == DT_F32)
    hipLaunchKernelGGL((colsum_kernel<float>), dim3(gx, gy), dim3(256), 0, stream, (const float*)A, lda, rows, cols, out, out2);
  else
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), dim3(gx, gy), dim3(256), 0, stream, (const bf16_t*)A, lda, rows, cols, out, out2);
  GIC_CHECK_LAUNCH("colsum");
  return GIC_OK;
}

int embedding_fwd(const float* weight, const int64_t* ids, float* out, long n, int V, int E, hipStream_t stream) {
  GIC_CHECK_ARG(weight && ids && out, "embedding_fwd: null pointer");
  if (n * E == 0) return GIC_OK;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(n * E)), dim3(256), 0, stream, weight, ids, out, n, V, E);
  GIC_CHECK_LAUNCH("embedding_fwd");
  return GIC_OK;
}

int embedding_bwd(const float* d_out, const int64_t* ids, float* d_weight, long n, int V, int E, int zero_first,
                  hipStream_t stream) {
  GIC_CHECK_ARG(d_out && ids && d_weight, "embedding_bwd: null pointer");
  if (zero_first) GIC_PROPAGATE(fill_zero(d_weight, (size_t)V * E * sizeof(float), stream));
  if (n * E == 0) return GIC_OK;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(n * E)), dim3(256), 0, stream, d_out, ids, d_weight, n, V, E);
  GIC_CHECK_LAUNCH("embedding_bwd");
  return GIC_OK;
}

}  // namespace gic

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int gic_abi_version(void) { return GIC_ABI_VERSION; }
const char* gic_last_error(void) { return gic::g_last_error; }

int gic_gemm(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
             int a_kc, int b_kc, int in_dtype, int out_dtype, const float* bias, int accumulate, float alpha,
             void* stream) {
  gic::GemmDesc d;
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
  d.a_kc = a_kc; d.b_kc = b_kc; d.in_dtype = in_dtype; d.out_dtype = out_dtype;
  d.bias = bias; d.accumulate = accumulate; d.alpha = alpha;
  return gic::gemm(d, (hipStream_t)stream);
}

int gic_cast2d(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd, int64_t rows,
               int64_t cols, void* stream) {
  return gic::cast2d(src, src_dtype, lds, dst, dst_dtype, ldd, rows, cols, (hipStream_t)stream);
}

int gic_embedding_fwd(const float* weight, const int64_t* ids, float* out, int64_t n, int32_t V, int32_t E, void* stream) {
  return gic::embedding_fwd(weight, ids, out, n, V, E, (hipStream_t)stream);
}

int gic_embedding_bwd(const float* d_out, const int64_t* ids, float* d_weight, int64_t n, int32_t V, int32_t E,
                      int zero_first, void* stream) {
  return gic::embedding_bwd(d_out, ids, d_weight, n, V, E, zero_first, (hipStream_t)stream);
}

}  // extern "C"
