// Image encoder (reference src/generator.py:8-25): ResNet trunk forward (no_grad, BatchNorm on batch
// statistics) + the trainable Linear/BatchNorm1d head.
//
// Activations are NHWC in the compute dtype, so a convolution is the implicit GEMM
//   out[(n,ho,wo), cout] = sum_{(r,s,c)} in[n, ho*st-pad+r, wo*st-pad+s, c] * w[cout, r, s, c]
// run by the MFMA GEMM kernel (gemm.hip, CONV loader) with the BatchNorm statistics (per-channel sum and
// sum of squares of the f32 accumulators) reduced in its epilogue.  Everything else here is memory-bound
// element-wise work with 16-byte accesses along the channel axis:
//   pack_image        NCHW f32 -> zero-bordered NHWC4 (3 channels + 1 zero) so the 7x7 stem is a plain
//                     [7 x 8 x 4] window with no bounds checks
//   bn_act            y -> relu(bn(y) + residual | bn(residual))
//   bn_relu_maxpool   stem: bn + relu + 3x3/2 max-pool in one pass
//   avgpool           global average pool
//   bn1d_fwd / bwd    the head's BatchNorm1d(momentum=0.01) on [B, E]
#include "../../include/gicap.h"
#include "kernels.h"
#include "bn_fold.h"
#include "conv_b2b.h"

namespace gic {
namespace {

constexpr float kBnEps = 1e-5f;

// ---- NCHW f32 [N,3,S,S] -> [N, S+2*pad, Wp, 4] act, zero border and zero 4th channel.  A thread packs PX adjacent pixels (one 16-byte
// store in bf16): unconditional loads from clamped coordinates (border pixels are zeroed by a select after the loads), 32-bit index math.
template <typename TA, int PX>
__global__ __launch_bounds__(256) void pack_image_kernel(const float* __restrict__ img, TA* __restrict__ out, int N, int S, int pad, int Hp, int Wp) {
  const int groups = Wp / PX;                                        // pixel groups per row (launch code: Wp % PX == 0)
  const int total = N * Hp * groups;                                 // < 2^31 (launch code)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int gq = i % groups, t = i / groups;
    const int hp = t % Hp, n = t / Hp;
    const int h = hp - pad;
    const bool hok = h >= 0 && h < S;
    const int hc = min(max(h, 0), S - 1);
    float v[PX][3];
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int w = gq * PX + q - pad;
      const int wc = min(max(w, 0), S - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[q][c] = img[(((long)n * 3 + c) * S + hc) * S + wc];
    }
    TA o[PX * 4];
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int w = gq * PX + q - pad;
      const bool ok = hok && w >= 0 && w < S;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[q * 4 + c] = from_f32<TA>(ok ? v[q][c] : 0.f);
      o[q * 4 + 3] = from_f32<TA>(0.f);
    }
    TA* dst = out + (long)i * (PX * 4);
    if constexpr (sizeof(TA) * PX * 4 == 16) *(uint4*)dst = *(const uint4*)o;
    else if constexpr (sizeof(TA) * PX * 4 == 8) *(uint2*)dst = *(const uint2*)o;
    else {
#pragma unroll
      for (int q = 0; q < PX; ++q) *(uint4*)(dst + q * 4) = *(const uint4*)(o + q * 4);      // f32: 16 bytes per pixel
    }
  }
}

// ---- conv weight [Cout, Cin, KH, KW] f32 -> [Cout, KH, KWp, Cinp] act (zero padded taps / channels)
template <typename TA>
__global__ void repack_conv_weight_kernel(const float* __restrict__ w, TA* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                          int Cinp, int KWp) {
  const long total = (long)Cout * KH * KWp * Cinp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cinp);
    long t = i / Cinp;
    const int s = (int)(t % KWp); t /= KWp;
    const int r = (int)(t % KH);
    const int co = (int)(t / KH);
    float v = 0.f;
    if (c < Cin && s < KW) v = w[(((long)co * Cin + c) * KH + r) * KW + s];
    out[i] = from_f32<TA>(v);
  }
}

struct BnSrc {            // y-side or residual-side BatchNorm inputs; stats == nullptr -> identity (no BN)
  const float* stats;     // [nrep][2C]: sum, sum of squares (replicas are summed here)
  int nrep;
  const float* gamma;
  const float* beta;
  const float* run_mean;  // eval mode (stats == nullptr but run_mean != nullptr)
  const float* run_var;
};

__device__ __forceinline__ void bn_coeffs(const BnSrc& b, int c, int C, float inv_count, float& scale, float& shift) {
  float mean, var;
  if (b.stats) {
    const float g = b.gamma[c], bt = b.beta[c];              // in flight together with the replicas (bn_fold.h)
    float s1, s2;
    fold_replicas(b.stats, b.nrep, C, c, s1, s2);
    mean = s1 * inv_count;
    var = fmaxf(s2 * inv_count - mean * mean, 0.f);
    scale = g * rsqrtf(var + kBnEps);
    shift = bt - mean * scale;
    return;
  } else if (b.run_mean) {
    mean = b.run_mean[c]; var = b.run_var[c];
  } else { scale = 1.f; shift = 0.f; return; }
  scale = b.gamma[c] * rsqrtf(var + kBnEps);
  shift = b.beta[c] - mean * scale;
}

// Block-cooperative BatchNorm coefficients: the 256 threads fold the statistics replicas of all C channels once
// into LDS ([scale | shift] (+ the residual's pair)); afterwards every element costs one LDS read per coefficient.
__device__ __forceinline__ void stage_coeffs(float* coef, const BnSrc& b, int C, float inv_count) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sc, sh;
    bn_coeffs(b, c, C, inv_count, sc, sh);
    coef[c] = sc;
    coef[C + c] = sh;
  }
}

template <typename TA> struct Vec16 { static constexpr int N = 16 / sizeof(TA); };

// ---- out = [relu]( bn(y) + (res ? bn_res(res) : 0) ), rows x C; 16 bytes (8 bf16 / 4 f32 channels) per access
template <typename TA, bool RES>
__global__ __launch_bounds__(1024) void bn_act_kernel(const TA* __restrict__ y, BnSrc by, const TA* __restrict__ res, BnSrc br,
                                                      float inv_count, int relu, TA* __restrict__ out, long rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float coef[];      // [4][C]
  constexpr int VN = Vec16<TA>::N;
  stage_coeffs(coef, by, C, inv_count);
  if (RES) stage_coeffs(coef + 2 * C, br, C, inv_count);
  __syncthreads();
  const int cv = C / VN;
  // gridDim.x*1024 is a multiple of cv (launch code), so a thread keeps one channel group for all its rows: its coefficients move
  // from LDS to registers once, its element offsets advance by a constant, and the loop is pure load / fma / store.
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(i0 % cv) * VN;
  float sc[VN], sh[VN], rs[VN], rh[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) {
    sc[k] = coef[c0 + k]; sh[k] = coef[C + c0 + k];
    rs[k] = RES ? coef[2 * C + c0 + k] : 0.f; rh[k] = RES ? coef[3 * C + c0 + k] : 0.f;
  }
  // 4 (+4) independent 16-byte loads in flight per thread (rows r, r+R, r+2R, r+3R) before any arithmetic; the loads are
  // unconditional (rows past the end re-read the thread's first row: nothing waits between them), the stores are not
  const long R = (long)gridDim.x * blockDim.x / cv;
  for (long r = i0 / cv; r < rows; r += 4 * R) {
    TA yv[4][VN], rv[4][VN];
    long o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long ru = r + u * R;
      o[u] = (ru < rows ? ru : r) * C + c0;
      *(uint4*)yv[u] = *(const uint4*)(y + o[u]);
      if (RES) *(uint4*)rv[u] = *(const uint4*)(res + o[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      TA ov[VN];
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        float v = to_f32<TA>(yv[u][k]) * sc[k] + sh[k];
        if (RES) v += to_f32<TA>(rv[u][k]) * rs[k] + rh[k];
        if (relu) v = fmaxf(v, 0.f);
        ov[k] = from_f32<TA>(v);
      }
      if (r + u * R < rows) *(uint4*)(out + o[u]) = *(const uint4*)ov;
    }
  }
}

// ---- stem: relu(bn(y)) then 3x3 stride-2 pad-1 max-pool.  y [N,H,W,C] -> out [N,Ho,Wo,C]
// The launch's thread count is a multiple of the channel-group count (launch code), so a thread keeps its channels: coefficients in
// registers; the nine window loads of an output are unconditional (a coordinate outside the image is clamped onto the window's own edge
// pixel: a duplicate inside a maximum changes nothing) and all in flight before the first use -- loads inside `if (inside)` made the
// wave wait for each tap in turn.
template <typename TA>
__global__ __launch_bounds__(1024) void bn_relu_maxpool_kernel(const TA* __restrict__ y, BnSrc by, float inv_count, TA* __restrict__ out,
                                                                int N, int H, int W, int C, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) float coef[];      // [2][C]
  constexpr int VN = Vec16<TA>::N;
  stage_coeffs(coef, by, C, inv_count);
  __syncthreads();
  const int cv = C / VN;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)(i0 % cv) * VN;
  float sc[VN], sh[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) { sc[k] = coef[c0 + k]; sh[k] = coef[C + c0 + k]; }
  const int pixels = N * Ho * Wo;                                    // < 2^31 (launch code)
  const int step = (int)((long)gridDim.x * blockDim.x / cv);
  for (int p = (int)(i0 / cv); p < pixels; p += step) {
    const int wo = p % Wo, t = p / Wo;
    const int ho = t % Ho, n = t / Ho;
    TA pv[9][VN];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = min(max(ho * 2 - 1 + r, 0), H - 1);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int w = min(max(wo * 2 - 1 + q, 0), W - 1);
        *(uint4*)pv[r * 3 + q] = *(const uint4*)(y + (((long)n * H + h) * W + w) * C + c0);
      }
    }
    float best[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) best[k] = 0.f;          // relu output is >= 0
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int k = 0; k < VN; ++k) best[k] = fmaxf(best[k], to_f32<TA>(pv[j][k]) * sc[k] + sh[k]);
    TA ov[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) ov[k] = from_f32<TA>(best[k]);
    *(uint4*)(out + (long)p * C + c0) = *(const uint4*)ov;
  }
}

// ---- global average pool: x [N, HW, C] -> out [N, C] (act) ; one thread per (n, c), 7 loads in flight
template <typename TA>
__global__ void avgpool_kernel(const TA* __restrict__ x, TA* __restrict__ out, int N, int HW, int C) {
  const long total = (long)N * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long n = i / C;
    const TA* xp = x + n * HW * C + c;
    float s = 0.f;
    int p = 0;
    for (; p + 7 <= HW; p += 7) {                    // sequential summation order kept (parity), loads issued together
      float v[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) v[q] = to_f32<TA>(xp[(long)(p + q) * C]);
#pragma unroll
      for (int q = 0; q < 7; ++q) s += v[q];
    }
    for (; p < HW; ++p) s += to_f32<TA>(xp[(long)p * C]);
    out[i] = from_f32<TA>(s / (float)HW);
  }
}

// ---- running statistics of every BatchNorm2d of the trunk in one launch (table built once by the host)
__global__ void bn_running_update_kernel(const gic_bn_running_desc* __restrict__ table, int nlayers) {
  const int l = blockIdx.x;
  if (l >= nlayers) return;
  const gic_bn_running_desc d = table[l];
  const float inv = 1.f / d.count;
  const float unbias = d.count > 1.f ? d.count / (d.count - 1.f) : 1.f;
  for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < d.C; c += blockDim.x * gridDim.y) {   // gridDim.y channel slabs: one round trip
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < d.nrep; ++r) { s1 += d.stats[(long)r * 2 * d.C + c]; s2 += d.stats[(long)r * 2 * d.C + d.C + c]; }
    const float mean = s1 * inv;
    const float var = fmaxf(s2 * inv - mean * mean, 0.f);
    d.running_mean[c] = (1.f - d.momentum) * d.running_mean[c] + d.momentum * mean;
    d.running_var[c] = (1.f - d.momentum) * d.running_var[c] + d.momentum * var * unbias;
  }
}

// ---- BatchNorm1d over the batch axis of x [B,E]: a block owns 16 feature columns, its 16 row groups stride the batch
// (16 loads in flight per column instead of one dependent chain) and fold through LDS
__device__ __forceinline__ float bn1d_colsum(float v, float (*red)[17], int r, int c) {
  red[r][c] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i][c];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ run_mean,
                                                        float* __restrict__ run_var, int training, float momentum, float eps,
                                                        float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ invstd,
                                                        int B, int E) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + c;
  const bool live = e < E;
  float mean = 0.f, var = 1.f;
  if (training) {
    float s = 0.f;
    if (live) for (int b = r; b < B; b += 16) s += x[(long)b * E + e];
    mean = bn1d_colsum(s, red, r, c) / (float)B;
    float q = 0.f;
    if (live) for (int b = r; b < B; b += 16) { const float d = x[(long)b * E + e] - mean; q += d * d; }
    var = bn1d_colsum(q, red, r, c) / (float)B;
    if (live && r == 0) {
      run_mean[e] = (1.f - momentum) * run_mean[e] + momentum * mean;
      run_var[e] = (1.f - momentum) * run_var[e] + momentum * var * (B > 1 ? (float)B / (float)(B - 1) : 1.f);
    }
  } else if (live) {
    mean = run_mean[e]; var = run_var[e];
  }
  if (!live) return;
  const float is = rsqrtf(var + eps);
  if (r == 0) invstd[e] = is;
  const float g = gamma[e], bt = beta[e];
  for (int b = r; b < B; b += 16) {
    const float xh = (x[(long)b * E + e] - mean) * is;
    xhat[(long)b * E + e] = xh;
    y[(long)b * E + e] = xh * g + bt;
  }
}

__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        int training, float* __restrict__ dx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int B, int E) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + c;
  const bool live = e < E;
  float sg = 0.f, sb = 0.f;
  if (live) for (int b = r; b < B; b += 16) { const float d = dy[(long)b * E + e]; sg += d * xhat[(long)b * E + e]; sb += d; }
  sg = bn1d_colsum(sg, red, r, c);
  sb = bn1d_colsum(sb, red, r, c);
  if (!live) return;
  if (r == 0) { dgamma[e] = sg; dbeta[e] = sb; }
  const float k = gamma[e] * invstd[e];
  const float invB = 1.f / (float)B;
  for (int b = r; b < B; b += 16) {
    const float d = dy[(long)b * E + e];
    dx[(long)b * E + e] = training ? k * (d - invB * (sb + xhat[(long)b * E + e] * sg)) : k * d;
  }
}

inline int grid1d(long total, int cap = 4096) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
// bn_act: 1024-thread blocks, at most 2 per CU: few blocks fold the statistics replicas (the per-block prologue), and
// gridDim.x*1024 stays a multiple of every power-of-two channel-group count <= 1024 (threads keep one channel group).
inline int bn_act_grid(long total) {
  long g = (total + 1023) / 1024;
  return (int)(g < 1 ? 1 : (g > 512 ? 512 : g));
}
// grid whose total thread count is a multiple of `period` work items (period = C/4, a power of two here)
inline int grid_periodic(long total, int period, int cap = 4096) {
  int g = grid1d(total, cap);
  const int mult = period > 256 ? period / 256 : 1;
  g = (g + mult - 1) / mult * mult;
  return g;
}

BnSrc make_src(const float* stats, int nrep, const float* gamma, const float* beta, const float* rm, const float* rv) {
  BnSrc s;
  s.stats = stats; s.nrep = nrep < 1 ? 1 : nrep; s.gamma = gamma; s.beta = beta; s.run_mean = rm; s.run_var = rv;
  return s;
}

}  // namespace
}  // namespace gic

using namespace gic;

extern "C" {

int gic_pack_image(const float* nchw, void* out, int dtype, int N, int S, int pad, int Wp, void* stream) {
  GIC_CHECK_ARG(nchw && out && N > 0 && S > 0 && pad >= 0 && Wp >= S + 2 * pad, "pack_image: bad argument");
  const int Hp = S + 2 * pad;
  const long total = (long)N * Hp * Wp;
  GIC_CHECK_ARG(total < (1l << 31) && (((uintptr_t)out) & 15) == 0, "pack_image: too many pixels or an unaligned output");
  const hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32) {
    hipLaunchKernelGGL((pack_image_kernel<float, 1>), dim3(grid1d(total, 8192)), dim3(256), 0, st, nchw, (float*)out, N, S, pad, Hp, Wp);
  } else if (Wp % 2 == 0) {
    hipLaunchKernelGGL((pack_image_kernel<bf16_t, 2>), dim3(grid1d(total / 2, 8192)), dim3(256), 0, st, nchw, (bf16_t*)out, N, S, pad, Hp, Wp);
  } else {
    hipLaunchKernelGGL((pack_image_kernel<bf16_t, 1>), dim3(grid1d(total, 8192)), dim3(256), 0, st, nchw, (bf16_t*)out, N, S, pad, Hp, Wp);
  }
  GIC_CHECK_LAUNCH("pack_image");
  return GIC_OK;
}

int gic_repack_conv_weight(const float* w, void* out, int dtype, int Cout, int Cin, int KH, int KW, int Cin_pad, int KW_pad, void* stream) {
  GIC_CHECK_ARG(w && out && Cin_pad >= Cin && KW_pad >= KW, "repack_conv_weight: bad argument");
  const long total = (long)Cout * KH * KW_pad * Cin_pad;
  if (dtype == DT_F32)
    hipLaunchKernelGGL((repack_conv_weight_kernel<float>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, w, (float*)out, Cout, Cin, KH, KW, Cin_pad, KW_pad);
  else
    hipLaunchKernelGGL((repack_conv_weight_kernel<bf16_t>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)out, Cout, Cin, KH, KW, Cin_pad, KW_pad);
  GIC_CHECK_LAUNCH("repack_conv_weight");
  return GIC_OK;
}

int gic_conv2d(const void* in, const void* w, void* out, float* stats, int stats_nrep, int dtype, int N, int H, int W, int Cin, int Cout,
               int KH, int KW, int stride, int pad, void* stream) {
  GIC_CHECK_ARG(in && w && out, "conv2d: null pointer");
  GIC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "conv2d: bad dims");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  GIC_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d: empty output");
  GemmDesc g;
  g.A = in; g.B = w; g.C = out;
  g.M = N * Ho * Wo; g.N = Cout; g.K = KH * KW * Cin;
  g.lda = Cin; g.ldb = g.K; g.ldc = Cout;
  g.in_dtype = dtype; g.out_dtype = dtype;
  g.conv = 1; g.cH = H; g.cW = W; g.cCin = Cin; g.cHo = Ho; g.cWo = Wo; g.cKH = KH; g.cKW = KW; g.cStride = stride; g.cPad = pad;
  g.epi = stats ? EPI_BNSTATS : EPI_PLAIN;
  g.stats = stats;
  g.stats_nrep = stats_nrep < 1 ? 1 : stats_nrep;
  return gemm(g, (hipStream_t)stream);
}

int gic_conv2d_bn_in(const void* in, const float* in_stats, int in_nrep, const float* in_gamma, const float* in_beta, float in_count,
                     const void* w, void* out, float* stats, int stats_nrep, int dtype, int N, int H, int W, int Cin, int Cout, int KH,
                     int KW, int stride, int pad, void* stream) {
  GIC_CHECK_ARG(in && in_stats && in_gamma && in_beta && w && out && stats && in_count > 0 && in_nrep >= 1, "conv2d_bn_in: bad argument");
  GIC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "conv2d_bn_in: bad dims");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  GIC_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d_bn_in: empty output");
  if (dtype != DT_BF16) return GIC_ERR_UNSUPPORTED;
  GemmDesc g;
  g.A = in; g.B = w; g.C = out;
  g.M = N * Ho * Wo; g.N = Cout; g.K = KH * KW * Cin;
  g.lda = Cin; g.ldb = g.K; g.ldc = Cout;
  g.in_dtype = dtype; g.out_dtype = dtype;
  g.conv = 1; g.cH = H; g.cW = W; g.cCin = Cin; g.cHo = Ho; g.cWo = Wo; g.cKH = KH; g.cKW = KW; g.cStride = stride; g.cPad = pad;
  g.epi = EPI_BNSTATS;
  g.stats = stats;
  g.stats_nrep = stats_nrep < 1 ? 1 : stats_nrep;
  g.in_stats = in_stats; g.in_nrep = in_nrep; g.in_gamma = in_gamma; g.in_beta = in_beta; g.in_inv_count = 1.f / in_count;
  return gemm(g, (hipStream_t)stream);
}

int gic_conv1x1_bn_in_stats(const void* in, const float* in_stats, int in_nrep, const float* in_gamma, const float* in_beta, float in_count,
                            const void* w, float* stats, int stats_nrep, int dtype, int64_t rows, int Cin, int Cout, void* stream) {
  GIC_CHECK_ARG(in && in_stats && in_gamma && in_beta && w && stats && in_count > 0 && in_nrep >= 1 && rows > 0 && Cin > 0 && Cout > 0,
                "conv1x1_bn_in_stats: bad argument");
  if (dtype != DT_BF16 || rows >= (1l << 31)) return GIC_ERR_UNSUPPORTED;
  GemmDesc g;
  g.A = in; g.B = w; g.C = (void*)in;                       // never written (stats_only): any non-null pointer
  g.M = (int)rows; g.N = Cout; g.K = Cin;
  g.lda = Cin; g.ldb = g.K; g.ldc = Cout;
  g.in_dtype = dtype; g.out_dtype = dtype;
  g.conv = 1; g.cH = 1; g.cW = (int)rows; g.cCin = Cin; g.cHo = 1; g.cWo = (int)rows; g.cKH = 1; g.cKW = 1; g.cStride = 1; g.cPad = 0;
  g.epi = EPI_BNSTATS;
  g.stats = stats;
  g.stats_nrep = stats_nrep < 1 ? 1 : stats_nrep;
  g.in_stats = in_stats; g.in_nrep = in_nrep; g.in_gamma = in_gamma; g.in_beta = in_beta; g.in_inv_count = 1.f / in_count;
  g.stats_only = 1;
  return gemm(g, (hipStream_t)stream);
}

int gic_conv_b2b(const void* y2, const float* stats2, int nrep2, const float* gamma2, const float* beta2, const void* w3, const float* stats3,
                 int nrep3, const float* gamma3, const float* beta3, const void* res, const float* res_stats, int res_nrep,
                 const float* res_gamma, const float* res_beta, float count, void* block_out, const void* w1n, void* y1n, float* stats1,
                 int nrep1, int dtype, int64_t rows, int C2, int C1N, void* stream) {
  GIC_CHECK_ARG(y2 && stats2 && gamma2 && beta2 && w3 && stats3 && gamma3 && beta3 && res && block_out && w1n && y1n && stats1 && count > 0 &&
                    nrep2 >= 1 && nrep3 >= 1 && rows > 0, "conv_b2b: bad argument");
  GIC_CHECK_ARG(!res_stats || (res_gamma && res_beta && res_nrep >= 1), "conv_b2b: the shortcut's BatchNorm needs gamma and beta");
  if (dtype != DT_BF16 || rows >= (1l << 31)) return GIC_ERR_UNSUPPORTED;
  B2bDesc d;
  d.y2 = y2; d.w3 = w3; d.res = res; d.w1n = w1n; d.out = block_out; d.y1n = y1n;
  d.stats2 = stats2; d.gamma2 = gamma2; d.beta2 = beta2; d.stats3 = stats3; d.gamma3 = gamma3; d.beta3 = beta3;
  d.res_stats = res_stats; d.res_gamma = res_gamma; d.res_beta = res_beta; d.stats1 = stats1;
  d.nrep2 = nrep2; d.nrep3 = nrep3; d.res_nrep = res_nrep; d.nrep1 = nrep1 < 1 ? 1 : nrep1;
  d.inv_count = 1.f / count; d.M = (int)rows;
  if (rows * C2 * 2 >= (1l << 31) || rows * 4 * C2 * 2 >= (1l << 31)) return GIC_ERR_UNSUPPORTED;
  d.y2_bytes = (unsigned)(rows * C2 * 2); d.res_bytes = (unsigned)(rows * 4 * C2 * 2);
  if (!try_conv_b2b(d, C2, C1N, (hipStream_t)stream)) return GIC_ERR_UNSUPPORTED;
  GIC_CHECK_LAUNCH("conv_b2b");
  return GIC_OK;
}

int gic_conv1x1_res_in(const void* in, const float* in_stats, int in_nrep, const float* in_gamma, const float* in_beta, const void* res,
                       const float* res_stats, int res_nrep, const float* res_gamma, const float* res_beta, float count, void* block_out,
                       const void* w, void* out, float* stats, int stats_nrep, int dtype, int N, int H, int W, int Cin, int Cout, void* stream) {
  GIC_CHECK_ARG(in && in_stats && in_gamma && in_beta && res && block_out && w && out && stats && count > 0 && in_nrep >= 1, "conv1x1_res_in: bad argument");
  GIC_CHECK_ARG(!res_stats || (res_gamma && res_beta && res_nrep >= 1), "conv1x1_res_in: the shortcut's BatchNorm needs gamma and beta");
  GIC_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv1x1_res_in: bad dims");
  if (dtype != DT_BF16) return GIC_ERR_UNSUPPORTED;
  GemmDesc g;
  g.A = in; g.B = w; g.C = out;
  g.M = N * H * W; g.N = Cout; g.K = Cin;
  g.lda = Cin; g.ldb = g.K; g.ldc = Cout;
  g.in_dtype = dtype; g.out_dtype = dtype;
  g.conv = 1; g.cH = H; g.cW = W; g.cCin = Cin; g.cHo = H; g.cWo = W; g.cKH = 1; g.cKW = 1; g.cStride = 1; g.cPad = 0;
  g.epi = EPI_BNSTATS;
  g.stats = stats;
  g.stats_nrep = stats_nrep < 1 ? 1 : stats_nrep;
  g.in_stats = in_stats; g.in_nrep = in_nrep; g.in_gamma = in_gamma; g.in_beta = in_beta; g.in_inv_count = 1.f / count;
  g.res = res; g.res_stats = res_stats; g.res_nrep = res_nrep; g.res_gamma = res_gamma; g.res_beta = res_beta; g.res_inv_count = 1.f / count;
  g.out_wb = block_out;
  return gemm(g, (hipStream_t)stream);
}

int gic_bn_act(const void* y, const float* stats, const float* gamma, const float* beta, const float* run_mean, const float* run_var,
               const void* res, const float* res_stats, const float* res_gamma, const float* res_beta, const float* res_run_mean,
               const float* res_run_var, int stats_nrep, float count, int relu, void* out, int dtype, int64_t rows, int C, void* stream) {
  GIC_CHECK_ARG(y && out && gamma && beta && (stats || run_mean) && rows > 0 && count > 0, "bn_act: bad argument");
  const BnSrc by = make_src(stats, stats_nrep, gamma, beta, run_mean, run_var);
  const BnSrc br = make_src(res_stats, stats_nrep, res_gamma, res_beta, res_run_mean, res_run_var);
  GIC_CHECK_ARG(C % 8 == 0 && (C & (C - 1)) == 0, "bn_act: C must be a power of two >= 8 (got %d)", C);
  const size_t lds = (size_t)4 * C * sizeof(float);
  const int cv = C / (dtype == DT_F32 ? 4 : 8);
  const long total = rows * cv;
  int grid = bn_act_grid(total);
  {  // threads keep their channel group: grid * 1024 must be a multiple of cv (it is for every power-of-two C)
    int step = cv, t = 1024;
    while (t) { const int m = step % t; step = t; t = m; }       // step = gcd(cv, 1024)
    const int mult = cv / step;
    grid = (grid + mult - 1) / mult * mult;
    // A thread takes rows r, r + R, r + 2R, r + 3R per pass (R = threads / cv); rows past the end re-read its first row.  With the
    // capped grid the last of the four is mostly such a duplicate (14x14 maps: R = 4096 against 12544 rows; 7x7: 2048 against 3136,
    // two of four).  Where ONE pass covers the rows within the cap, size the grid so that 4 R just does: every load useful
    // (block outputs of the 14x14 / 7x7 maps: 18.2 -> 16.6 us; with more passes per thread the smaller grid loses: 13.5 -> 14.4 us).
    static const bool fit = getenv("GIC_BN_ACT_NO_FIT") == nullptr;
    if (fit && total > 4 * 1024) {
      const long rr = (rows + 3) / 4;                             // rows per stride
      long g = (rr * cv + 1023) / 1024;
      g = (g + mult - 1) / mult * mult;
      if (g <= 512) grid = (int)g;
    }
  }
  const hipStream_t st = (hipStream_t)stream;
  if (dtype == DT_F32) {
    if (res) hipLaunchKernelGGL((bn_act_kernel<float, true>), dim3(grid), dim3(1024), lds, st, (const float*)y, by, (const float*)res, br, 1.f / count, relu, (float*)out, (long)rows, C);
    else hipLaunchKernelGGL((bn_act_kernel<float, false>), dim3(grid), dim3(1024), lds, st, (const float*)y, by, (const float*)res, br, 1.f / count, relu, (float*)out, (long)rows, C);
  } else {
    if (res) hipLaunchKernelGGL((bn_act_kernel<bf16_t, true>), dim3(grid), dim3(1024), lds, st, (const bf16_t*)y, by, (const bf16_t*)res, br, 1.f / count, relu, (bf16_t*)out, (long)rows, C);
    else hipLaunchKernelGGL((bn_act_kernel<bf16_t, false>), dim3(grid), dim3(1024), lds, st, (const bf16_t*)y, by, (const bf16_t*)res, br, 1.f / count, relu, (bf16_t*)out, (long)rows, C);
  }
  GIC_CHECK_LAUNCH("bn_act");
  return GIC_OK;
}

int gic_bn_relu_maxpool(const void* y, const float* stats, const float* gamma, const float* beta, const float* run_mean,
                        const float* run_var, int stats_nrep, float count, void* out, int dtype, int N, int H, int W, int C, void* stream) {
  GIC_CHECK_ARG(y && out && gamma && beta && (stats || run_mean) && C % 8 == 0, "bn_relu_maxpool: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const BnSrc by = make_src(stats, stats_nrep, gamma, beta, run_mean, run_var);
  const size_t lds = (size_t)2 * C * sizeof(float);
  const int cv = C / (dtype == DT_F32 ? 4 : 8);
  const long total = (long)N * Ho * Wo * cv;
  GIC_CHECK_ARG((long)N * Ho * Wo < (1l << 31), "bn_relu_maxpool: too many output pixels");
  // 1024-thread blocks (few prologues); grid * 1024 a multiple of the channel-group count: threads keep their channels
  int grid = (int)((total + 1023) / 1024);
  grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
  {
    int step = cv, t = 1024;
    while (t) { const int m = step % t; step = t; t = m; }       // gcd(cv, 1024)
    const int mult = cv / step;
    grid = (grid + mult - 1) / mult * mult;
  }
  if (dtype == DT_F32)
    hipLaunchKernelGGL((bn_relu_maxpool_kernel<float>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, (const float*)y, by, 1.f / count, (float*)out, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL((bn_relu_maxpool_kernel<bf16_t>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, (const bf16_t*)y, by, 1.f / count, (bf16_t*)out, N, H, W, C, Ho, Wo);
  GIC_CHECK_LAUNCH("bn_relu_maxpool");
  return GIC_OK;
}

int gic_avgpool(const void* x, void* out, int dtype, int N, int HW, int C, void* stream) {
  GIC_CHECK_ARG(x && out && N > 0 && HW > 0 && C > 0, "avgpool: bad argument");
  const long total = (long)N * C;
  if (dtype == DT_F32)
    hipLaunchKernelGGL((avgpool_kernel<float>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, N, HW, C);
  else
    hipLaunchKernelGGL((avgpool_kernel<bf16_t>), dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, N, HW, C);
  GIC_CHECK_LAUNCH("avgpool");
  return GIC_OK;
}

int gic_bn_running_update(const gic_bn_running_desc* table_dev, int nlayers, void* stream) {
  GIC_CHECK_ARG(table_dev && nlayers > 0, "bn_running_update: bad argument");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(nlayers, 8), dim3(256), 0, (hipStream_t)stream, table_dev, nlayers);
  GIC_CHECK_LAUNCH("bn_running_update");
  return GIC_OK;
}

int gic_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int training,
                 float momentum, float eps, float* y, float* xhat, float* invstd, int B, int E, void* stream) {
  GIC_CHECK_ARG(x && gamma && beta && running_mean && running_var && y && xhat && invstd && B > 0 && E > 0, "bn1d_fwd: bad argument");
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cdiv(E, 16)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, running_mean, running_var,
                     training, momentum, eps, y, xhat, invstd, B, E);
  GIC_CHECK_LAUNCH("bn1d_fwd");
  return GIC_OK;
}

int gic_bn1d_bwd(const float* dy, const float* xhat, const float* invstd, const float* gamma, int training, float* dx, float* dgamma,
                 float* dbeta, int B, int E, void* stream) {
  GIC_CHECK_ARG(dy && xhat && invstd && gamma && dx && dgamma && dbeta && B > 0 && E > 0, "bn1d_bwd: bad argument");
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cdiv(E, 16)), dim3(256), 0, (hipStream_t)stream, dy, xhat, invstd, gamma, training, dx,
                     dgamma, dbeta, B, E);
  GIC_CHECK_LAUNCH("bn1d_bwd");
  return GIC_OK;
}

int gic_colsum(const void* A, int dtype, int64_t lda, int64_t rows, int64_t cols, float* out, int accumulate, void* stream) {
  return colsum(A, dtype, lda, rows, cols, out, nullptr, accumulate, (hipStream_t)stream);
}

}  // extern "C"
