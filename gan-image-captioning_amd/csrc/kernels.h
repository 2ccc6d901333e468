// Internal host-side launchers shared between translation units (all enqueue on `stream`).
#pragma once
#include "common.h"
#include "gemm.h"

namespace gic {

// dst[r*ldd + c] = cast(src[r*lds + c])
int cast2d(const void* src, int src_dtype, long lds, void* dst, int dst_dtype, long ldd, long rows, long cols,
           hipStream_t stream);
// out[c] (+)= sum_r A[r*lda + c]   (A in `dtype`, out f32; out2 optional second destination)
int colsum(const void* A, int dtype, long lda, long rows, long cols, float* out, float* out2, int accumulate,
           hipStream_t stream);
int embedding_fwd(const float* weight, const int64_t* ids, float* out, long n, int V, int E, hipStream_t stream);
int embedding_bwd(const float* d_out, const int64_t* ids, float* d_weight, long n, int V, int E, int zero_first,
                  hipStream_t stream);
int fill_zero(void* p, size_t bytes, hipStream_t stream);
// dst[c][r] = src[r][c] for a row-major [rows, cols] matrix of `dtype`
int transpose2d(const void* src, void* dst, int dtype, long rows, long cols, hipStream_t stream);

// decoder.hip internals shared with attention.hip
int decoder_output_bwd(int dt, int B, int L, int V, int H, const void* probs, const void* d_out, float temperature, const float* t_dev, int pretrain,
                       void* dlogits_ws, const void* wout, const void* hout, float* dhout, float* d_wout, float* d_bout, hipStream_t stream);
int embed_scatter_time(const float* dx, long ld, const int64_t* ids, float* d_embed, int B, int L, int E, int V, hipStream_t stream, long ids_stride = 0);

}  // namespace gic
