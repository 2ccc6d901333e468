// Sequence discriminator: Discriminator.forward / backward (reference src/discriminator.py:34-62).
//
//   emb[B*L,De] = inp W_emb^T (MFMA GEMM, or a gather when the input is token ids)
//   pooled[B*R,F] = max_t relu(conv_f(emb))   fused conv+bias+ReLU+max-over-time: the
//                   [B,300,L-f+1,64] pre-pool tensor of the reference is never materialised;
//                   one block per (b, representation) stages its L x s window in LDS
//   highway       = GEMM with fused gate + dropout epilogue (EPI_HIGHWAY)
//   head          = GEMM 900->100, then a 100-wide dot per row
// Activations [B*R, Fp] keep a zero-padded leading dim Fp (multiple of 8) so the bf16 GEMMs
// can use 16-byte chunks along K; pad columns are zero (written by these kernels or, for
// `ydrop`, left from a zero-initialised allocation - the GEMM epilogue never touches them).
#include <stdlib.h>
#include "../../include/gicap.h"
#include "kernels.h"

namespace gic {
namespace {

constexpr int kMaxTaps = 32;     // f * s per filter held in registers
constexpr int kOutDim = 100;     // feature2out width (discriminator.py:28)
constexpr int kOutPad = 104;

struct ConvMeta {
  int nconv, F, Fp, s;
  int fsize[GIC_MAX_CONVS], nfilt[GIC_MAX_CONVS], foff[GIC_MAX_CONVS];
  const float* w[GIC_MAX_CONVS];
  const float* b[GIC_MAX_CONVS];
  float* dw[GIC_MAX_CONVS];
  float* db[GIC_MAX_CONVS];
};

__device__ __forceinline__ int conv_of(const ConvMeta& cm, int col) {
  int k = 0;
  while (k + 1 < cm.nconv && col >= cm.foff[k + 1]) ++k;
  return k;
}

// ---- ids input: emb[(b,l), e] = W_emb[e, id]   (one_hot(real) @ W^T as a gather; training.py:158)
__global__ void disc_emb_gather_kernel(const float* __restrict__ w, const int64_t* __restrict__ ids, float* __restrict__ emb,
                                       long rows, int De, int V) {
  const long total = rows * De;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / De;
    const int e = (int)(i % De);
    long id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    emb[i] = w[(long)e * V + id];
  }
}
__global__ void disc_emb_scatter_kernel(const float* __restrict__ demb_f32, const void* __restrict__ demb, int dt,
                                        const int64_t* __restrict__ ids, float* __restrict__ dw, long rows, int De, int V) {
  const long total = rows * De;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / De;
    const int e = (int)(i % De);
    long id = ids[r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    atomicAdd(&dw[(long)e * V + id], ld_as_f32(demb, i, dt));
  }
}

// One filter of width F over a single embedding column (s == 1): sliding register window, one LDS read and F fmas per
// time step, relu then running max with first-index tie-break (discriminator.py:42,45).
template <int F>
__device__ __forceinline__ void conv_max_s1(const float* __restrict__ xs, const float* __restrict__ wp, float bias, int L,
                                            float& best, int& bt) {
  float w[F], win[F];
#pragma unroll
  for (int j = 0; j < F; ++j) w[j] = wp[j];
#pragma unroll
  for (int j = 1; j < F; ++j) win[j] = xs[j - 1];
  best = -1.f; bt = 0;
  for (int t = 0; t + F <= L; ++t) {
#pragma unroll
    for (int j = 0; j + 1 < F; ++j) win[j] = win[j + 1];
    win[F - 1] = xs[t + F - 1];
    float v = bias;
#pragma unroll
    for (int j = 0; j < F; ++j) v += w[j] * win[j];
    v = fmaxf(v, 0.f);
    if (v > best) { best = v; bt = t; }
  }
}

// ---- fused conv + bias + ReLU + max-over-time.  grid = B*R blocks, 256 threads stride over the F filters.
template <typename TA, int MAXT>
__global__ __launch_bounds__(256) void disc_conv_pool_fwd_kernel(const float* __restrict__ emb, ConvMeta cm, int L, int De, int R,
                                                                   TA* __restrict__ pooled, uint8_t* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [L][s]
  const int br = blockIdx.x, b = br / R, r = br % R, s = cm.s;
  for (int i = threadIdx.x; i < L * s; i += 256) xs[i] = emb[((long)b * L + i / s) * De + r * s + i % s];
  __syncthreads();
  for (int col = threadIdx.x; col < cm.Fp; col += 256) {
    float best = 0.f;
    int bt = 0;
    if (col < cm.F) {
      const int k = conv_of(cm, col);
      const int f = cm.fsize[k], ch = col - cm.foff[k], taps = f * s;
      const float bias = cm.b[k][ch];
      const float* wp = cm.w[k] + (long)ch * taps;
      bool done = false;
      if (s == 1) {
        done = true;
        switch (f) {
          case 1: conv_max_s1<1>(xs, wp, bias, L, best, bt); break;
          case 2: conv_max_s1<2>(xs, wp, bias, L, best, bt); break;
          case 3: conv_max_s1<3>(xs, wp, bias, L, best, bt); break;
          case 4: conv_max_s1<4>(xs, wp, bias, L, best, bt); break;
          case 5: conv_max_s1<5>(xs, wp, bias, L, best, bt); break;
          case 6: conv_max_s1<6>(xs, wp, bias, L, best, bt); break;
          case 7: conv_max_s1<7>(xs, wp, bias, L, best, bt); break;
          case 8: conv_max_s1<8>(xs, wp, bias, L, best, bt); break;
          default: done = false;
        }
      }
      if (!done) {
        float w[MAXT];
#pragma unroll
        for (int j = 0; j < MAXT; ++j) w[j] = j < taps ? wp[j] : 0.f;
        best = -1.f;
        for (int t = 0; t + f <= L; ++t) {
          float v = bias;
#pragma unroll
          for (int j = 0; j < MAXT; ++j)
            if (j < taps) v += w[j] * xs[t * s + j];      // window (t..t+f-1) x s is contiguous in xs
          v = fmaxf(v, 0.f);                              // relu then max (discriminator.py:42,45)
          if (v > best) { best = v; bt = t; }
        }
      }
    }
    pooled[(long)br * cm.Fp + col] = from_f32<TA>(best);
    if (argmax) argmax[(long)br * cm.Fp + col] = (uint8_t)bt;
  }
}

// ---- the same on the matrix cores (s == 1, filter widths <= 8): v_mfma_f32_16x16x4_f32 -- fp32 products and accumulation, so both
// compute modes keep the fp32 convolution of the reference.  One workgroup = 16 (caption, representation) pairs; per filter tile (16
// filters of one width) and time step t one MFMA (two for widths 5..8) forms D[pair][filter] = bias + sum_j x[pair][t + j] w[filter][j]:
//   A[pair][k] = x[pair][t + k]   (lane: pair = lane & 15, k = lane >> 4; one LDS read)
//   B[k][filter] = w[filter][k]   (lane: filter = lane & 15, k = lane >> 4; held in registers across t)
//   C = bias[filter]              (every row of the accumulator tile)
// then relu, running max and first-index argmax per (pair, filter) in registers (discriminator.py:42,45).  The scalar kernel above
// re-reads every filter's weights per pair (57 MB of L2 traffic per launch at cfg2) and reaches 14 TFLOP/s: 35 us for 0.5 GFLOP.
template <typename TA>
__global__ __launch_bounds__(512) void disc_conv_pool_fwd_mfma_kernel(const float* __restrict__ emb, ConvMeta cm, int L, int De, int R, long rowsBR,
                                                                        TA* __restrict__ pooled, uint8_t* __restrict__ argmax) {
  __shared__ float xs[16][269];                                  // [pair][t], zero beyond L (L <= 255, + the taps and the 4-step blocks' overhang; odd stride: banks)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const long br0 = (long)blockIdx.x * 16;
  for (int i = tid; i < 16 * 268; i += 512) {
    const int p = i / 268, t = i - p * 268;
    const long br = br0 + p;
    float v = 0.f;
    if (br < rowsBR && t < L) v = emb[((br / R) * L + t) * De + (br % R)];       // s == 1: representation r reads embedding column r
    xs[p][t] = v;
  }
  __syncthreads();
  // filter tiles: for each width k, ceil(nfilt / 16) tiles of 16 filters; waves take them round robin
  int tile = 0;
  for (int k = 0; k < cm.nconv; ++k) {
    const int f = cm.fsize[k], nf = cm.nfilt[k], T = L - f + 1;
    const float* wk = cm.w[k];
    const float* bk = cm.b[k];
    for (int c0 = 0; c0 < nf; c0 += 16, ++tile) {
      if ((tile & 7) != w) continue;                             // wave-uniform
      const int ch = c0 + li;
      const bool ok = ch < nf;
      // B fragments: taps lk and 4 + lk of filter ch (zero beyond the width / the filter count)
      const float b0 = (ok && lk < f) ? wk[(long)ch * f + lk] : 0.f;
      const float b1 = (ok && 4 + lk < f) ? wk[(long)ch * f + 4 + lk] : 0.f;
      const float bias = ok ? bk[ch] : 0.f;
      // relu then max over time with the first index on ties (discriminator.py:42,45) = a running maximum that starts at (0, t = 0)
      // and moves on strictly greater pre-activations only: one compare and two selects per element.  The epilogue is VALU work
      // of the size of the MFMAs it follows, so four time steps go together (independent accumulators under the previous ones'
      // selects; steps past T read the zero padding and are skipped).
      float best[4] = {0.f, 0.f, 0.f, 0.f};
      int bt[4] = {0, 0, 0, 0};
      for (int t0 = 0; t0 < T; t0 += 4) {
        f32x4 acc[4];
        float xa[4], xb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xa[u] = xs[li][t0 + u + lk]; xb[u] = xs[li][t0 + u + 4 + lk]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u] = (f32x4){bias, bias, bias, bias};
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u], b0, acc[u], 0, 0, 0);
        }
        if (f > 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[u], b1, acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (t0 + u < T) {                                      // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (acc[u][r] > best[r]) { best[r] = acc[u][r]; bt[r] = t0 + u; }
          }
        }
      }
      // accumulator row 4 lk + r = pair, column li = filter
      if (ok) {
        const int col = cm.foff[k] + ch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long br = br0 + 4 * lk + r;
          if (br < rowsBR) {
            pooled[br * cm.Fp + col] = from_f32<TA>(best[r]);
            if (argmax) argmax[br * cm.Fp + col] = (uint8_t)bt[r];
          }
        }
      }
    }
  }
  // pad columns F .. Fp-1 (zero, argmax 0: as the scalar kernel writes them)
  for (int i = tid; i < 16 * (cm.Fp - cm.F); i += 512) {
    const int p = i / (cm.Fp - cm.F), c = cm.F + i % (cm.Fp - cm.F);
    const long br = br0 + p;
    if (br < rowsBR) { pooled[br * cm.Fp + c] = from_f32<TA>(0.f); if (argmax) argmax[br * cm.Fp + c] = 0; }
  }
}

// ---- the forward-only form in the bf16 mode (reward evaluation of the SeqGAN-style step: no backward pass follows, no argmax is kept):
// bf16 products on v_mfma_f32_32x32x16_bf16.  The fp32 kernel above is bound by the fp32 matrix pipe (one 16x16x4 product per 16 pairs x
// 16 filters and time step: 123 of its 157 TFLOP/s); here one instruction forms 32 pairs x 32 filters of a time step with K = 16 >= the
// filter width (rows of B beyond the width are zero, so the A window may run on into the next time steps), and the epilogue is one v_max
// per element.  One workgroup = 32 (caption, representation) pairs: their embedding columns as bf16 windows xw[pair][t][0..7] =
// x[pair][t .. t + 7] in LDS (one aligned 16-byte read per A fragment).  Values differ from the fp32 kernel's by bf16 rounding of the
// embedding and the filter weights (the reward path already runs its highway layer and head in bf16).
template <typename TA>
__global__ __launch_bounds__(512) void disc_conv_pool_fwd_bf16_kernel(const float* __restrict__ emb, ConvMeta cm, int L, int De, int R, long rowsBR,
                                                                        TA* __restrict__ pooled) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  // dynamic LDS: the windows xw [32 pairs][L][8] bf16, then the fp32 staging xs [32][TP] with TP = L + 9 (8 of zero overhang, odd pitch)
  extern __shared__ __attribute__((aligned(16))) unsigned char cp_smem[];
  const int TP = L + 9;
  bf16_t* xwp = (bf16_t*)cp_smem;
  float* xsp = (float*)(cp_smem + (size_t)32 * L * 16);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long br0 = (long)blockIdx.x * 32;
  for (int i = tid; i < 32 * TP; i += 512) {
    const int p = i / TP, t = i - p * TP;
    const long br = br0 + p;
    float v = 0.f;
    if (br < rowsBR && t < L) v = emb[((br / R) * L + t) * De + (br % R)];       // s == 1: representation r reads embedding column r
    xsp[p * TP + t] = v;
  }
  __syncthreads();
  for (int i = tid; i < 32 * L; i += 512) {
    const int p = i / L, t = i - p * L;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)xsp[p * TP + t + e];
    *(bf16x8*)(xwp + ((long)p * L + t) * 8) = v;
  }
  __syncthreads();
  int tile = 0;
  for (int k = 0; k < cm.nconv; ++k) {
    const int f = cm.fsize[k], nf = cm.nfilt[k], T = L - f + 1;
    const float* wk = cm.w[k];
    const float* bk = cm.b[k];
    for (int c0 = 0; c0 < nf; c0 += 32, ++tile) {
      if ((tile & 7) != w) continue;                             // wave-uniform
      const int ch = c0 + li;
      const bool ok = ch < nf;
      // B fragment: lane (filter li, k half lh) holds w[filter][8 lh .. 8 lh + 7]: the taps in the lower half, zero beyond the width
      bf16x8 fb;
#pragma unroll
      for (int e = 0; e < 8; ++e) fb[e] = (bf16_t)((ok && lh == 0 && e < f) ? wk[(long)ch * f + e] : 0.f);
      const float bias = ok ? bk[ch] : 0.f;
      f32x16 best;
#pragma unroll
      for (int i = 0; i < 16; ++i) best[i] = 0.f;                // relu, then max over time: a running maximum that starts at 0
      for (int t0 = 0; t0 < T; t0 += 2) {                        // two time steps together: independent accumulators
        f32x16 a0, a1;
#pragma unroll
        for (int i = 0; i < 16; ++i) a0[i] = a1[i] = bias;
        const bf16x8 fa0 = *(const bf16x8*)(xwp + ((long)li * L + t0) * 8);
        const bf16x8 fa1 = *(const bf16x8*)(xwp + ((long)li * L + (t0 + 1 < T ? t0 + 1 : t0)) * 8);   // (an odd T repeats its last step: the maximum is unchanged)
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, a1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) best[i] = fmaxf(best[i], fmaxf(a0[i], a1[i]));
      }
      // accumulator register i: pair 8 (i / 4) + 4 lh + (i % 4), column li = filter
      if (ok) {
        const int col = cm.foff[k] + ch;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const long br = br0 + 8 * (i >> 2) + 4 * lh + (i & 3);
          if (br < rowsBR) pooled[br * cm.Fp + col] = from_f32<TA>(best[i]);
        }
      }
    }
  }
  // pad columns F .. Fp-1 (zero, as the other kernels write them)
  for (int i = tid; i < 32 * (cm.Fp - cm.F); i += 512) {
    const int p = i / (cm.Fp - cm.F), c = cm.F + i % (cm.Fp - cm.F);
    const long br = br0 + p;
    if (br < rowsBR) pooled[br * cm.Fp + c] = from_f32<TA>(0.f);
  }
}

// ---- conv backward, input side: d emb.  One block per (b, r) owns its s embedding columns.
template <typename TA>
__global__ __launch_bounds__(256) void disc_conv_pool_bwd_x_kernel(const float* __restrict__ dpooled, const TA* __restrict__ pooled,
                                                                     const uint8_t* __restrict__ argmax, ConvMeta cm, int L, int De, int R,
                                                                     TA* __restrict__ demb) {
  // Deterministic (no atomics): stage the gated gradient and argmax of all F filters in LDS, then each
  // of the L*s outputs is summed by NG threads over disjoint filter subsets in a fixed order.
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int br = blockIdx.x, b = br / R, r = br % R, s = cm.s;
  const int n_out = L * s;
  const int NG = n_out >= 256 ? 1 : 256 / n_out;
  float* geff = lds_f;                               // [Fp]
  float* part = lds_f + cm.Fp;                       // [NG][n_out]
  int* tst = (int*)(part + NG * n_out);              // [Fp]  argmax time
  int* meta = tst + cm.Fp;                           // [Fp]  (taps << 20) | offset of the filter's weights in wl
  float* wl = (float*)(meta + cm.Fp);                // all conv weights, filter-major
  {
    int woff = 0;
    for (int k = 0; k < cm.nconv; ++k) {
      const int taps = cm.fsize[k] * s, nw = cm.nfilt[k] * taps;
      for (int i = threadIdx.x; i < nw; i += 256) wl[woff + i] = cm.w[k][i];
      for (int ch = threadIdx.x; ch < cm.nfilt[k]; ch += 256) meta[cm.foff[k] + ch] = (taps << 20) | (woff + ch * taps);
      woff += nw;
    }
  }
  for (int col = threadIdx.x; col < cm.F; col += 256) {
    const long o = (long)br * cm.Fp + col;
    geff[col] = to_f32<TA>(pooled[o]) > 0.f ? dpooled[o] : 0.f;     // relu gate
    tst[col] = argmax[o];
  }
  __syncthreads();
  for (int item = threadIdx.x; item < NG * n_out; item += 256) {
    const int o = item % n_out, grp = item / n_out;
    const int t = o / s, e = o % s;
    float acc = 0.f;
    for (int col = grp; col < cm.F; col += NG) {
      const float g = geff[col];
      const int mt = meta[col];
      const int dtap = (t - tst[col]) * s + e;       // tap index of this output inside the filter's window
      if (g != 0.f && dtap >= e && dtap < (mt >> 20)) acc += g * wl[(mt & 0xFFFFF) + dtap];
    }
    part[grp * n_out + o] = acc;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < n_out; o += 256) {
    float acc = 0.f;
    for (int g = 0; g < NG; ++g) acc += part[g * n_out + o];
    demb[((long)b * L + o / s) * De + r * s + o % s] = from_f32<TA>(acc);
  }
}

// Same result for few outputs per row (n_out = L*s <= MAXO, e.g. s = 1) and short windows (taps <= MAXT): a thread owns F/256
// filters with their weights in registers and scatters g*w into ITS OWN column of an LDS sheet part[o][thread] (no other
// thread touches that column: plain adds, no atomics, bank = thread id -> conflict-free); the sheet is then folded per output
// in a fixed order (deterministic).  A block serves `rpb` rows with the same registers.
template <typename TA, int MAXO, int MAXT>
__global__ __launch_bounds__(256) void disc_conv_pool_bwd_x_small_kernel(const float* __restrict__ dpooled, const TA* __restrict__ pooled,
                                                                           const uint8_t* __restrict__ argmax, ConvMeta cm, int L, int De,
                                                                           int R, TA* __restrict__ demb, int rows, int rpb) {
  extern __shared__ __attribute__((aligned(16))) float part_raw[];     // [n_out][256] (n_out <= MAXO): 20 KB at L = 20, eight workgroups per CU
  float (*part)[256] = (float (*)[256])part_raw;
  const int s = cm.s, n_out = L * s;
  constexpr int FPT = 4;                             // filters per thread (F <= 1024)
  int f_taps[FPT];
  float f_w[FPT][MAXT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int col = threadIdx.x + i * 256;
    f_taps[i] = 0;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) f_w[i][j] = 0.f;
    if (col < cm.F) {
      const int k = conv_of(cm, col);
      f_taps[i] = cm.fsize[k] * s;
      const float* wp = cm.w[k] + (long)(col - cm.foff[k]) * f_taps[i];
#pragma unroll
      for (int j = 0; j < MAXT; ++j) if (j < f_taps[i]) f_w[i][j] = wp[j];
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int br = blockIdx.x * rpb; br < (blockIdx.x + 1) * rpb && br < rows; ++br) {
    const int b = br / R, r = br % R;
    // this pair's gated gradients and window starts, all twelve loads in flight at once (clamped columns, masked afterwards: loads under
    // a per-filter branch were waited for one by one).  (LDS float atomics instead of the read-modify-write below: 133 us against 47.)
    float g[FPT];
    int t0[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      const int col = threadIdx.x + i * 256;
      const long idx = (long)br * cm.Fp + (col < cm.Fp ? col : cm.Fp - 1);
      const float p = to_f32<TA>(pooled[idx]), dg = dpooled[idx];
      t0[i] = (int)argmax[idx] * s;                  // output index of the window's first tap
      g[i] = (col < cm.F && p > 0.f) ? dg : 0.f;     // relu gate
    }
    for (int o = 0; o < n_out; ++o) part[o][threadIdx.x] = 0.f;
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
#pragma unroll
      for (int j = 0; j < MAXT; ++j)
        if (j < f_taps[i] && t0[i] + j < n_out) part[t0[i] + j][threadIdx.x] += g[i] * f_w[i][j];
    }
    __syncthreads();
    for (int o = w; o < n_out; o += 4) {             // wave w folds outputs w, w+4, ...: 4 columns per lane, then the wave
      const float v = wave_sum(part[o][lane] + part[o][lane + 64] + part[o][lane + 128] + part[o][lane + 192]);
      if (lane == 0) demb[((long)b * L + o / s) * De + r * s + o % s] = from_f32<TA>(v);
    }
    __syncthreads();
  }
}

// ---- conv backward, weight side: block = 64 filters x 4 row lanes, rows split over gridDim.y
template <typename TA, int MAXT>
__global__ __launch_bounds__(256) void disc_conv_pool_bwd_w_kernel(const float* __restrict__ dpooled, const TA* __restrict__ pooled,
                                                                     const uint8_t* __restrict__ argmax, const float* __restrict__ emb,
                                                                     ConvMeta cm, int L, int De, int R, long rows) {
  __shared__ float red[4][64][MAXT + 1];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6, s = cm.s;
  const int col = blockIdx.x * 64 + cx;
  float acc[MAXT + 1];
#pragma unroll
  for (int j = 0; j <= MAXT; ++j) acc[j] = 0.f;
  int k = 0, taps = 0, ch = 0;
  if (col < cm.F) {
    k = conv_of(cm, col);
    taps = cm.fsize[k] * s;
    ch = col - cm.foff[k];
    for (long m = (long)blockIdx.y * 4 + ry; m < rows; m += (long)gridDim.y * 4) {
      const long o = m * cm.Fp + col;
      if (to_f32<TA>(pooled[o]) <= 0.f) continue;
      const float g = dpooled[o];
      const int b = (int)(m / R), r = (int)(m % R), t = argmax[o];
      const float* x = emb + ((long)b * L + t) * De + r * s;     // tap (j, e) at x[j*De + e]
#pragma unroll
      for (int j = 0; j < MAXT; ++j)
        if (j < taps) acc[j] += g * x[(j / s) * De + (j % s)];
      acc[MAXT] += g;
    }
  }
#pragma unroll
  for (int j = 0; j <= MAXT; ++j) red[ry][cx][j] = acc[j];
  __syncthreads();
  if (ry == 0 && col < cm.F) {
    for (int j = 0; j < taps; ++j)
      atomicAdd(&cm.dw[k][(long)ch * taps + j], red[0][cx][j] + red[1][cx][j] + red[2][cx][j] + red[3][cx][j]);
    atomicAdd(&cm.db[k][ch], red[0][cx][MAXT] + red[1][cx][MAXT] + red[2][cx][MAXT] + red[3][cx][MAXT]);
  }
}

// The same for s == 1 with the embedding columns of one caption staged in LDS as xs[representation][t]: the eight taps of a
// (pair, filter) are then LDS reads of one 20-float row (lanes = filters of one pair: broadcast / few banks) instead of eight
// scattered 4-byte global loads at stride De (one address per lane per load kept the texture path busy for 82 us at cfg2; the
// streams pooled / dpooled / argmax are 55 MB): 27 us.  block = 64 filters x 4 waves over the representations, captions over
// gridDim.y.  (Four filters per thread -- a quarter of the load instructions, 16-byte loads -- took 62-74 us: the kernel lives on
// the number of waves in flight, not on the load count.)
template <typename TA, int MAXT>
__global__ __launch_bounds__(256) void disc_conv_pool_bwd_w_lds_kernel(const float* __restrict__ dpooled, const TA* __restrict__ pooled,
                                                                         const uint8_t* __restrict__ argmax, const float* __restrict__ emb,
                                                                         ConvMeta cm, int L, int De, int R, int ncap) {
  extern __shared__ __attribute__((aligned(16))) float xs[];       // [R][LP], LP = L + MAXT | 1 (zero beyond L)
  __shared__ float red[4][64][MAXT + 1];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int LP = (L + MAXT) | 1;
  const int col = blockIdx.x * 64 + cx;
  const bool okc = col < cm.F;
  const int colc = okc ? col : 0;
  float acc[MAXT + 1];
#pragma unroll
  for (int j = 0; j <= MAXT; ++j) acc[j] = 0.f;
  const int k = conv_of(cm, colc), taps = cm.fsize[k], ch = colc - cm.foff[k];
  for (int i = threadIdx.x; i < R * LP; i += 256) xs[i] = 0.f;
  for (int b = blockIdx.y; b < ncap; b += gridDim.y) {
    __syncthreads();
    for (int i = threadIdx.x; i < R * L; i += 256) {
      const int t = i / R, r = i - t * R;
      xs[r * LP + t] = emb[((long)b * L + t) * De + r];
    }
    __syncthreads();
    // eight pairs per pass: their 24 loads leave together (clamped rows, masked afterwards), then the LDS windows.  The relu gate is a
    // 0 / 1 factor: a select lets the compiler sink the dpooled load under the pooled test (two dependent round trips per pair).
    constexpr int U = 8;
    for (int r0 = ry; r0 < R; r0 += 4 * U) {
      float pv[U], dg[U];
      int tt[U], rc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = r0 + 4 * u;
        rc[u] = r < R ? r : R - 1;
        const long o = ((long)b * R + rc[u]) * cm.Fp + colc;
        pv[u] = to_f32<TA>(pooled[o]);
        dg[u] = dpooled[o];
        tt[u] = argmax[o];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float g = dg[u] * ((okc && r0 + 4 * u < R && pv[u] > 0.f) ? 1.f : 0.f);
        const float* x = xs + rc[u] * LP + tt[u];                    // the window starts at the argmax (<= L - width)
#pragma unroll
        for (int j = 0; j < MAXT; ++j) acc[j] += g * x[j];           // taps beyond the width: dropped when the sums leave
        acc[MAXT] += g;
      }
    }
  }
#pragma unroll
  for (int j = 0; j <= MAXT; ++j) red[ry][cx][j] = acc[j];
  __syncthreads();
  if (ry == 0 && okc) {
    for (int j = 0; j < taps; ++j)
      atomicAdd(&cm.dw[k][(long)ch * taps + j], red[0][cx][j] + red[1][cx][j] + red[2][cx][j] + red[3][cx][j]);
    atomicAdd(&cm.db[k][ch], red[0][cx][MAXT] + red[1][cx][MAXT] + red[2][cx][MAXT] + red[3][cx][MAXT]);
  }
}

// ---- logits[m] = feat[m,:] . w + b   (out2logits, discriminator.py:60)
__global__ void disc_out_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ bias,
                                    float* __restrict__ logits, long rows) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= rows) return;
  float s = bias[0];
  for (int j = 0; j < kOutDim; ++j) s += feat[m * kOutDim + j] * w[j];
  logits[m] = s;
}

// dfeat[m,j] = dlogit[m] * w[j] (pad cols zero); optional dw[j] += sum_m dlogit[m] feat[m,j], db += sum_m dlogit[m]
template <typename TA>
__global__ __launch_bounds__(256) void disc_out_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat,
                                                             const float* __restrict__ w, TA* __restrict__ dfeat,
                                                             float* __restrict__ dw, float* __restrict__ db, long rows) {
  __shared__ float red[2][128];
  const int j = threadIdx.x & 127, half = threadIdx.x >> 7;
  float aw = 0.f, ab = 0.f;
  const float wj = j < kOutDim ? w[j] : 0.f;
  for (long m = (long)blockIdx.x * 2 + half; m < rows; m += (long)gridDim.x * 2) {
    const float g = dlogits[m];
    if (j < kOutPad) dfeat[m * kOutPad + j] = from_f32<TA>(g * wj);
    if (dw && j < kOutDim) aw += g * feat[m * kOutDim + j];
    if (j == 0) ab += g;
  }
  if (!dw) return;
  red[half][j] = aw;
  __syncthreads();
  if (half == 0 && j < kOutDim) atomicAdd(&dw[j], red[0][j] + red[1][j]);
  __syncthreads();
  red[half][j] = ab;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(db, red[0][0] + red[1][0]);
}

// ---- highway backward pointwise (discriminator.py:53-58 differentiated):
//   dy = dydrop * keep * scale;  dh = dy * (sg(1-sg)(relu(h)-x) + sg*[h>0]);  dx_direct = dy * (1-sg)
template <typename TA>
__global__ void disc_highway_bwd_kernel(const float* __restrict__ dydrop, const uint8_t* __restrict__ keep, float scale,
                                        const float* __restrict__ hpre, const TA* __restrict__ pooled, TA* __restrict__ dh,
                                        float* __restrict__ dpooled, long rows, int F, int Fp) {
  const long total = rows * Fp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Fp);
    float vdh = 0.f, vdx = 0.f;
    if (n < F) {
      float dy = dydrop[i] * scale;
      if (keep) dy *= (float)keep[i];
      const float h = hpre[i], x = to_f32<TA>(pooled[i]);
      const float sg = 1.f / (1.f + expf(-h));
      vdh = dy * (sg * (1.f - sg) * (fmaxf(h, 0.f) - x) + (h > 0.f ? sg : 0.f));
      vdx = dy * (1.f - sg);
    }
    dh[i] = from_f32<TA>(vdh);
    dpooled[i] = vdx;
  }
}

// The same, 8 consecutive columns per thread (Fp % 8 == 0, 16-byte aligned buffers): 16-byte accesses and one 32-bit division per 8
// elements instead of 4-byte accesses and a 64-bit modulo per element (cfg2, in-step: 25 -> 19 us per launch).
template <typename TA>
__global__ __launch_bounds__(256) void disc_highway_bwd_vec_kernel(const float* __restrict__ dydrop, const uint8_t* __restrict__ keep, float scale,
                                                                     const float* __restrict__ hpre, const TA* __restrict__ pooled,
                                                                     TA* __restrict__ dh, float* __restrict__ dpooled, unsigned groups, int F, int G8) {
  for (unsigned g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
    const int n0 = (int)(g % (unsigned)G8) * 8;
    const long i = (long)g * 8;
    const float4 d0 = *(const float4*)(dydrop + i), d1 = *(const float4*)(dydrop + i + 4);
    const float4 h0 = *(const float4*)(hpre + i), h1 = *(const float4*)(hpre + i + 4);
    const float dy8[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const float h8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    unsigned long long kb = 0x0101010101010101ull;
    if (keep) kb = *(const unsigned long long*)(keep + i);
    __attribute__((aligned(16))) TA x8[8], o8[8];
#pragma unroll
    for (int q = 0; q < (int)(8 * sizeof(TA) / 16); ++q) ((float4*)x8)[q] = ((const float4*)(pooled + i))[q];
    float dx8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float vdh = 0.f, vdx = 0.f;
      if (n0 + e < F) {
        const float dy = dy8[e] * scale * (float)((kb >> (8 * e)) & 0xffull);
        const float h = h8[e], x = to_f32<TA>(x8[e]);
        const float sg = 1.f / (1.f + expf(-h));
        vdh = dy * (sg * (1.f - sg) * (fmaxf(h, 0.f) - x) + (h > 0.f ? sg : 0.f));
        vdx = dy * (1.f - sg);
      }
      o8[e] = from_f32<TA>(vdh);
      dx8[e] = vdx;
    }
#pragma unroll
    for (int q = 0; q < (int)(8 * sizeof(TA) / 16); ++q) ((float4*)(dh + i))[q] = ((const float4*)o8)[q];
    *(float4*)(dpooled + i) = make_float4(dx8[0], dx8[1], dx8[2], dx8[3]);
    *(float4*)(dpooled + i + 4) = make_float4(dx8[4], dx8[5], dx8[6], dx8[7]);
  }
}

struct DCtx {
  int B, L, V, De, R, s, F, Fp, dt;
  float drop_p;              // nn.Dropout p of discriminator.py:10,30
  long rowsBL, rowsBR;
  float keep_scale(int train) const { return train ? 1.f / (1.f - drop_p) : 1.f; }
  ConvMeta cm;
};

int make_ctx(const gic_disc_dims* d, const gic_disc_params* P, const gic_disc_grads* G, DCtx& c) {
  GIC_CHECK_ARG(d, "disc: null dims");
  GIC_CHECK_ARG(d->B > 0 && d->L > 0 && d->V > 0 && d->De > 0 && d->R > 0, "disc: bad dims");
  GIC_CHECK_ARG(d->De % d->R == 0, "disc: disc_embed_dim %d not divisible by disc_num_rep %d", d->De, d->R);
  GIC_CHECK_ARG(d->nconv >= 1 && d->nconv <= GIC_MAX_CONVS, "disc: nconv must be 1..%d", GIC_MAX_CONVS);
  GIC_CHECK_ARG(d->dtype == DT_F32 || d->dtype == DT_BF16, "disc: bad dtype");
  c.B = d->B; c.L = d->L; c.V = d->V; c.De = d->De; c.R = d->R; c.s = d->De / d->R; c.dt = d->dtype;
  c.F = d->F; c.Fp = d->Fp;
  GIC_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f, "disc: dropout p=%g must be in [0, 1)", (double)d->drop_p);
  c.drop_p = d->drop_p;
  c.rowsBL = (long)d->B * d->L; c.rowsBR = (long)d->B * d->R;
  GIC_CHECK_ARG(c.Fp >= c.F && c.Fp % 8 == 0, "disc: Fp=%d must be >= F=%d and a multiple of 8", c.Fp, c.F);
  GIC_CHECK_ARG(c.L <= 255, "disc: L=%d exceeds the uint8 argmax range", c.L);
  ConvMeta& cm = c.cm;
  cm.nconv = d->nconv; cm.F = c.F; cm.Fp = c.Fp; cm.s = c.s;
  int off = 0;
  for (int k = 0; k < d->nconv; ++k) {
    GIC_CHECK_ARG(d->fsize[k] >= 1 && d->fsize[k] <= d->L, "disc: filter size %d does not fit caption length %d", d->fsize[k], d->L);
    GIC_CHECK_ARG(d->fsize[k] * c.s <= kMaxTaps, "disc: filter taps f*s=%d exceed %d", d->fsize[k] * c.s, kMaxTaps);
    cm.fsize[k] = d->fsize[k]; cm.nfilt[k] = d->nfilt[k]; cm.foff[k] = off; off += d->nfilt[k];
    cm.w[k] = P ? P->conv_w[k] : nullptr; cm.b[k] = P ? P->conv_b[k] : nullptr;
    cm.dw[k] = G ? G->conv_w[k] : nullptr; cm.db[k] = G ? G->conv_b[k] : nullptr;
    GIC_CHECK_ARG(!P || (cm.w[k] && cm.b[k]), "disc: null conv %d parameter", k);
  }
  GIC_CHECK_ARG(off == c.F, "disc: F=%d != sum(nfilt)=%d", c.F, off);
  return GIC_OK;
}

inline int grid1d(long total, int cap = 2048) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// feature2out + out2logits on st->ydrop
int disc_head_fwd(const DCtx& c, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* st, float* logits,
                  hipStream_t stream) {
  GemmDesc g;
  g.A = st->ydrop; g.lda = c.Fp; g.B = S->f2o_w; g.ldb = c.Fp; g.C = st->feat; g.ldc = kOutDim;
  g.M = (int)c.rowsBR; g.N = kOutDim; g.K = c.Fp; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.bias = P->f2o_b;
  GIC_PROPAGATE(gemm(g, stream));
  hipLaunchKernelGGL(disc_out_fwd_kernel, dim3(cdiv(c.rowsBR, 256)), dim3(256), 0, stream, (const float*)st->feat, P->o2l_w, P->o2l_b,
                     logits, c.rowsBR);
  GIC_CHECK_LAUNCH("disc_out_fwd");
  return GIC_OK;
}

template <typename TA>
int disc_fwd_t(const DCtx& c, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* st,
               const void* inp_soft, long ld_inp, const int64_t* inp_ids, int train, const uint8_t* keep_mask,
               uint64_t seed, const uint64_t* seed_dev, float* logits, hipStream_t stream) {
  // 1. embedding
  if (inp_ids) {
    hipLaunchKernelGGL(disc_emb_gather_kernel, dim3(grid1d(c.rowsBL * c.De)), dim3(256), 0, stream, P->emb, inp_ids, st->emb,
                       c.rowsBL, c.De, c.V);
    GIC_CHECK_LAUNCH("disc_emb_gather");
  } else {
    GemmDesc g;
    g.A = inp_soft; g.lda = ld_inp; g.B = S->emb; g.ldb = c.V; g.C = st->emb; g.ldc = c.De;
    g.M = (int)c.rowsBL; g.N = c.De; g.K = c.V; g.in_dtype = c.dt; g.out_dtype = DT_F32;
    GIC_PROPAGATE(gemm(g, stream));
  }
  // 2. conv + relu + max over time
  int max_taps = 0;
  for (int k = 0; k < c.cm.nconv; ++k) max_taps = c.cm.fsize[k] * c.s > max_taps ? c.cm.fsize[k] * c.s : max_taps;
  static const bool no_mfma = getenv("GIC_NO_DISC_CONV_MFMA") != nullptr;
  static const bool no_bf16 = getenv("GIC_NO_DISC_CONV_BF16") != nullptr;
  if (c.s == 1 && max_taps <= 8 && !no_mfma && !no_bf16 && sizeof(TA) == 2 && !st->argmax && c.L <= 64)
    hipLaunchKernelGGL((disc_conv_pool_fwd_bf16_kernel<TA>), dim3((unsigned)((c.rowsBR + 31) / 32)), dim3(512),
                       (size_t)32 * c.L * 16 + (size_t)32 * (c.L + 9) * 4, stream,
                       (const float*)st->emb, c.cm, c.L, c.De, c.R, c.rowsBR, (TA*)st->pooled);
  else if (c.s == 1 && max_taps <= 8 && !no_mfma)
    hipLaunchKernelGGL((disc_conv_pool_fwd_mfma_kernel<TA>), dim3((unsigned)((c.rowsBR + 15) / 16)), dim3(512), 0, stream,
                       (const float*)st->emb, c.cm, c.L, c.De, c.R, c.rowsBR, (TA*)st->pooled, st->argmax);
  else if (max_taps <= 8)
    hipLaunchKernelGGL((disc_conv_pool_fwd_kernel<TA, 8>), dim3((unsigned)c.rowsBR), dim3(256), c.L * c.s * sizeof(float), stream,
                       (const float*)st->emb, c.cm, c.L, c.De, c.R, (TA*)st->pooled, st->argmax);
  else
    hipLaunchKernelGGL((disc_conv_pool_fwd_kernel<TA, kMaxTaps>), dim3((unsigned)c.rowsBR), dim3(256), c.L * c.s * sizeof(float), stream,
                       (const float*)st->emb, c.cm, c.L, c.De, c.R, (TA*)st->pooled, st->argmax);
  GIC_CHECK_LAUNCH("disc_conv_pool_fwd");
  // 3. highway + dropout (fused epilogue)
  {
    GemmDesc g;
    g.A = st->pooled; g.lda = c.Fp; g.B = S->hw_w; g.ldb = c.Fp; g.C = st->ydrop; g.ldc = c.Fp;
    g.M = (int)c.rowsBR; g.N = c.F; g.K = c.Fp; g.in_dtype = c.dt; g.out_dtype = c.dt; g.bias = P->hw_b;
    g.epi = EPI_HIGHWAY; g.X = st->pooled; g.ldx = c.Fp; g.Hpre = st->hpre; g.ldh = c.Fp;
    if (train) {
      g.keep_scale = c.keep_scale(1);        // nn.Dropout(p), discriminator.py:10,30
      g.drop_p = c.drop_p;
      if (keep_mask) { g.mask = keep_mask; g.ldmask = c.F; } else { g.use_philox = 1; g.seed = seed; g.seed_dev = seed_dev; g.stream = 0x44495343ull; }
      g.mask_out = st->keep; g.ldmask_out = c.Fp;
    }
    GIC_PROPAGATE(gemm(g, stream));
  }
  return disc_head_fwd(c, P, S, st, logits, stream);
}

// The same highway output under ANOTHER dropout draw, from the saved pre-activation h and carry x (no GEMM):
// y = sig(h) relu(h) + (1 - sig(h)) x, exactly the EPI_HIGHWAY epilogue incl. its Philox indexing (4 rows per draw).
template <typename TA>
__global__ void disc_highway_redrop_kernel(const float* __restrict__ hpre, const TA* __restrict__ pooled,
                                           const uint8_t* __restrict__ mask, int train, uint64_t seed_val, const uint64_t* __restrict__ seed_dev, float drop_p,
                                           float keep_scale, TA* __restrict__ ydrop, uint8_t* __restrict__ keep_out,
                                           long rows, int F, int Fp) {
  const long groups = (rows + 3) / 4;
  const uint64_t seed = seed_dev ? *seed_dev : seed_val;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups * F; i += (long)gridDim.x * blockDim.x) {
    const long m4 = i / F;
    const int n = (int)(i % F);
    float keep4[4] = {1.f, 1.f, 1.f, 1.f};
    if (train && !mask) {
      uint32_t r0, r1, r2, r3;
      Philox::gen4(seed, 0x44495343ull, (uint64_t)m4 * (uint64_t)F + (uint64_t)n, r0, r1, r2, r3);
      keep4[0] = Philox::u01(r0) >= drop_p ? 1.f : 0.f;
      keep4[1] = Philox::u01(r1) >= drop_p ? 1.f : 0.f;
      keep4[2] = Philox::u01(r2) >= drop_p ? 1.f : 0.f;
      keep4[3] = Philox::u01(r3) >= drop_p ? 1.f : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m4 * 4 + r;
      if (m >= rows) break;
      const float h = hpre[m * Fp + n];
      const float x = to_f32<TA>(pooled[m * Fp + n]);
      const float sg = 1.f / (1.f + expf(-h));
      const float y = sg * fmaxf(h, 0.f) + (1.f - sg) * x;
      float keep = keep4[r];
      if (train && mask) keep = (float)mask[m * F + n];
      if (keep_out) keep_out[m * Fp + n] = (uint8_t)keep;
      ydrop[m * Fp + n] = from_f32<TA>(y * keep * keep_scale);
    }
  }
}

template <typename TA>
int disc_fwd_redrop_t(const DCtx& c, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* src,
                      const gic_disc_state* dst, int train, const uint8_t* keep_mask, uint64_t seed, const uint64_t* seed_dev, float* logits,
                      hipStream_t stream) {
  const long groups = (c.rowsBR + 3) / 4;
  hipLaunchKernelGGL((disc_highway_redrop_kernel<TA>), dim3(grid1d(groups * c.F)), dim3(256), 0, stream, (const float*)src->hpre,
                     (const TA*)src->pooled, keep_mask, train, seed, seed_dev, c.drop_p, c.keep_scale(train), (TA*)dst->ydrop,
                     train ? dst->keep : nullptr, c.rowsBR, c.F, c.Fp);
  GIC_CHECK_LAUNCH("disc_highway_redrop");
  return disc_head_fwd(c, P, S, dst, logits, stream);
}

template <typename TA>
int disc_bwd_t(const DCtx& c, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* st,
               const gic_disc_bwd_ws* ws, const void* inp_soft, long ld_inp, const int64_t* inp_ids, int train,
               const float* d_logits, const gic_disc_grads* G, int accumulate, void* d_inp, long ld_dinp, hipStream_t stream) {
  const long MR = c.rowsBR;
  if (G && !accumulate) {   // atomically-accumulated small grads start from zero
    GIC_PROPAGATE(fill_zero(G->o2l_w, kOutDim * sizeof(float), stream));
    GIC_PROPAGATE(fill_zero(G->o2l_b, sizeof(float), stream));
    for (int k = 0; k < c.cm.nconv; ++k) {
      GIC_PROPAGATE(fill_zero(G->conv_w[k], (size_t)c.cm.nfilt[k] * c.cm.fsize[k] * c.s * sizeof(float), stream));
      GIC_PROPAGATE(fill_zero(G->conv_b[k], (size_t)c.cm.nfilt[k] * sizeof(float), stream));
    }
    if (inp_ids) GIC_PROPAGATE(fill_zero(G->emb, (size_t)c.De * c.V * sizeof(float), stream));
  }
  // 1. out2logits backward
  hipLaunchKernelGGL((disc_out_bwd_kernel<TA>), dim3(256), dim3(256), 0, stream, d_logits, (const float*)st->feat, P->o2l_w,
                     (TA*)ws->dfeat, G ? G->o2l_w : nullptr, G ? G->o2l_b : nullptr, MR);
  GIC_CHECK_LAUNCH("disc_out_bwd");
  // 2. feature2out backward
  {
    GemmDesc g;   // dydrop[MR, Fp] = dfeat[MR,104] f2o_w[104, Fp]
    g.A = ws->dfeat; g.lda = kOutPad; g.a_kc = 1; g.B = S->f2o_w; g.ldb = c.Fp; g.b_kc = 0; g.C = ws->dydrop; g.ldc = c.Fp;
    g.M = (int)MR; g.N = c.Fp; g.K = kOutPad; g.in_dtype = c.dt; g.out_dtype = DT_F32;
    GIC_PROPAGATE(gemm(g, stream));
    if (G) {
      GemmDesc w;   // dW_f2o[100, F] = dfeat^T ydrop
      w.A = ws->dfeat; w.lda = kOutPad; w.a_kc = 0; w.B = st->ydrop; w.ldb = c.Fp; w.b_kc = 0; w.C = G->f2o_w; w.ldc = c.F;
      w.M = kOutDim; w.N = c.F; w.K = (int)MR; w.in_dtype = c.dt; w.out_dtype = DT_F32; w.accumulate = accumulate;
      GIC_PROPAGATE(gemm(w, stream));
      GIC_PROPAGATE(colsum(ws->dfeat, c.dt, kOutPad, MR, kOutDim, G->f2o_b, nullptr, accumulate, stream));
    }
  }
  // 3. highway backward
  const bool hb_vec = MR * c.Fp / 8 < (1l << 31) && ((((uintptr_t)ws->dydrop) | ((uintptr_t)st->hpre) | ((uintptr_t)st->pooled) | ((uintptr_t)ws->dh) |
                                                     ((uintptr_t)ws->dpooled)) & 15) == 0 && (!train || (((uintptr_t)st->keep) & 7) == 0);
  if (hb_vec)
    hipLaunchKernelGGL((disc_highway_bwd_vec_kernel<TA>), dim3(grid1d(MR * c.Fp / 8)), dim3(256), 0, stream, (const float*)ws->dydrop,
                       train ? (const uint8_t*)st->keep : nullptr, c.keep_scale(train), (const float*)st->hpre,
                       (const TA*)st->pooled, (TA*)ws->dh, ws->dpooled, (unsigned)(MR * c.Fp / 8), c.F, c.Fp / 8);
  else
    hipLaunchKernelGGL((disc_highway_bwd_kernel<TA>), dim3(grid1d(MR * c.Fp)), dim3(256), 0, stream, (const float*)ws->dydrop,
                       train ? (const uint8_t*)st->keep : nullptr, c.keep_scale(train), (const float*)st->hpre,
                       (const TA*)st->pooled, (TA*)ws->dh, ws->dpooled, MR, c.F, c.Fp);
  GIC_CHECK_LAUNCH("disc_highway_bwd");
  {
    GemmDesc g;   // dpooled += dh hw_w
    g.A = ws->dh; g.lda = c.Fp; g.a_kc = 1; g.B = S->hw_w_t ? S->hw_w_t : S->hw_w; g.ldb = c.Fp; g.b_kc = S->hw_w_t ? 1 : 0;
    g.C = ws->dpooled; g.ldc = c.Fp;
    g.M = (int)MR; g.N = c.Fp; g.K = c.Fp; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.accumulate = 1;
    GIC_PROPAGATE(gemm(g, stream));
    if (G) {
      GemmDesc w;   // dW_hw[F,F] = dh^T pooled
      w.A = ws->dh; w.lda = c.Fp; w.a_kc = 0; w.B = st->pooled; w.ldb = c.Fp; w.b_kc = 0; w.C = G->hw_w; w.ldc = c.F;
      w.M = c.F; w.N = c.F; w.K = (int)MR; w.in_dtype = c.dt; w.out_dtype = DT_F32; w.accumulate = accumulate;
      GIC_PROPAGATE(gemm(w, stream));
      GIC_PROPAGATE(colsum(ws->dh, c.dt, c.Fp, MR, c.F, G->hw_b, nullptr, accumulate, stream));
    }
  }
  // 4. conv / pool backward
  if (G) {
    int gy = cdiv(MR, 4 * 8);
    gy = gy > 64 ? 64 : gy;
    int mt = 0;
    for (int k = 0; k < c.cm.nconv; ++k) mt = c.cm.fsize[k] * c.s > mt ? c.cm.fsize[k] * c.s : mt;
    static const bool no_lds = getenv("GIC_NO_DISC_BWDW_LDS") != nullptr;
    if (mt <= 8 && c.s == 1 && c.R <= 64 && c.R <= c.De && MR % c.R == 0 && c.L <= 64 && !no_lds) {
      const int ncap = (int)(MR / c.R);
      hipLaunchKernelGGL((disc_conv_pool_bwd_w_lds_kernel<TA, 8>), dim3(cdiv(c.F, 64), ncap < 64 ? ncap : 64), dim3(256),
                         (size_t)c.R * ((c.L + 8) | 1) * sizeof(float), stream, (const float*)ws->dpooled, (const TA*)st->pooled,
                         (const uint8_t*)st->argmax, (const float*)st->emb, c.cm, c.L, c.De, c.R, ncap);
    } else if (mt <= 8)
      hipLaunchKernelGGL((disc_conv_pool_bwd_w_kernel<TA, 8>), dim3(cdiv(c.F, 64), gy), dim3(256), 0, stream, (const float*)ws->dpooled,
                         (const TA*)st->pooled, (const uint8_t*)st->argmax, (const float*)st->emb, c.cm, c.L, c.De, c.R, MR);
    else
      hipLaunchKernelGGL((disc_conv_pool_bwd_w_kernel<TA, kMaxTaps>), dim3(cdiv(c.F, 64), gy), dim3(256), 0, stream, (const float*)ws->dpooled,
                         (const TA*)st->pooled, (const uint8_t*)st->argmax, (const float*)st->emb, c.cm, c.L, c.De, c.R, MR);
    GIC_CHECK_LAUNCH("disc_conv_pool_bwd_w");
  }
  const int n_out = c.L * c.s;
  size_t conv_w_total = 0;
  for (int k = 0; k < c.cm.nconv; ++k) conv_w_total += (size_t)c.cm.nfilt[k] * c.cm.fsize[k] * c.s;
  const size_t bwd_x_lds = ((size_t)3 * c.Fp + (size_t)(n_out >= 256 ? 1 : 256 / n_out) * n_out + conv_w_total) * sizeof(float);
  GIC_CHECK_ARG(bwd_x_lds <= 160 * 1024, "disc_bwd: conv weights (%zu floats) do not fit the LDS staging", conv_w_total);
  int max_taps_x = 0;
  for (int k = 0; k < c.cm.nconv; ++k) max_taps_x = c.cm.fsize[k] * c.s > max_taps_x ? c.cm.fsize[k] * c.s : max_taps_x;
  if (n_out <= 32 && c.F <= 1024 && max_taps_x <= 8) {
    static const int rpb_env = [] { const char* e = getenv("GIC_BWDX_RPB"); return e ? atoi(e) : 0; }();
    const int rpb = rpb_env > 0 ? rpb_env : (MR >= 2048 ? 4 : 1);                // rows per block
    hipLaunchKernelGGL((disc_conv_pool_bwd_x_small_kernel<TA, 32, 8>), dim3((unsigned)cdiv(MR, rpb)), dim3(256), (size_t)n_out * 256 * sizeof(float), stream,
                       (const float*)ws->dpooled, (const TA*)st->pooled, (const uint8_t*)st->argmax, c.cm, c.L, c.De, c.R,
                       (TA*)ws->demb, (int)MR, rpb);
  } else {
    hipLaunchKernelGGL((disc_conv_pool_bwd_x_kernel<TA>), dim3((unsigned)MR), dim3(256), bwd_x_lds, stream,
                       (const float*)ws->dpooled, (const TA*)st->pooled, (const uint8_t*)st->argmax, c.cm, c.L, c.De, c.R, (TA*)ws->demb);
  }
  GIC_CHECK_LAUNCH("disc_conv_pool_bwd_x");
  // 5. embedding backward
  if (G) {
    if (inp_ids && inp_soft) {
      // mixed batch (one backward for the step's real and fake passes): the first half of the captions came in as token ids,
      // the second half as soft rows; the gathered half scatters first (G->emb zeroed above or accumulated into)
      const long half = c.rowsBL / 2;
      hipLaunchKernelGGL(disc_emb_scatter_kernel, dim3(grid1d(half * c.De)), dim3(256), 0, stream, (const float*)nullptr,
                         (const void*)ws->demb, c.dt, inp_ids, G->emb, half, c.De, c.V);
      GIC_CHECK_LAUNCH("disc_emb_scatter");
      GemmDesc w;   // dW_emb[De, V] += demb[half:]^T inp
      w.A = (const char*)ws->demb + (size_t)half * c.De * dtype_size(c.dt); w.lda = c.De; w.a_kc = 0; w.B = inp_soft; w.ldb = ld_inp; w.b_kc = 0;
      w.C = G->emb; w.ldc = c.V;
      w.M = c.De; w.N = c.V; w.K = (int)half; w.in_dtype = c.dt; w.out_dtype = DT_F32; w.accumulate = 1;
      GIC_PROPAGATE(gemm(w, stream));
    } else if (inp_ids) {
      hipLaunchKernelGGL(disc_emb_scatter_kernel, dim3(grid1d(c.rowsBL * c.De)), dim3(256), 0, stream, (const float*)nullptr,
                         (const void*)ws->demb, c.dt, inp_ids, G->emb, c.rowsBL, c.De, c.V);
      GIC_CHECK_LAUNCH("disc_emb_scatter");
    } else {
      GemmDesc w;   // dW_emb[De, V] = demb^T inp
      w.A = ws->demb; w.lda = c.De; w.a_kc = 0; w.B = inp_soft; w.ldb = ld_inp; w.b_kc = 0; w.C = G->emb; w.ldc = c.V;
      w.M = c.De; w.N = c.V; w.K = (int)c.rowsBL; w.in_dtype = c.dt; w.out_dtype = DT_F32; w.accumulate = accumulate;
      GIC_PROPAGATE(gemm(w, stream));
    }
  }
  if (d_inp) {
    GemmDesc g;   // d_inp[B*L, V] = demb[B*L, De] W_emb[De, V]
    g.A = ws->demb; g.lda = c.De; g.a_kc = 1; g.B = S->emb; g.ldb = c.V; g.b_kc = 0; g.C = d_inp; g.ldc = ld_dinp;
    g.M = (int)c.rowsBL; g.N = c.V; g.K = c.De; g.in_dtype = c.dt; g.out_dtype = c.dt;
    GIC_PROPAGATE(gemm(g, stream));
  }
  return GIC_OK;
}

}  // namespace
}  // namespace gic

using namespace gic;

extern "C" {

int gic_disc_state_bytes(const gic_disc_dims* dims, uint64_t* out) {
  DCtx c;
  GIC_PROPAGATE(make_ctx(dims, nullptr, nullptr, c));
  GIC_CHECK_ARG(out, "disc_state_bytes: null out");
  const uint64_t a = dtype_size(c.dt), MR = c.rowsBR, Fp = c.Fp;
  out[0] = (uint64_t)c.rowsBL * c.De * 4;      // emb
  out[1] = MR * Fp * a;                        // pooled
  out[2] = MR * Fp;                            // argmax
  out[3] = MR * Fp * 4;                        // hpre
  out[4] = MR * Fp;                            // keep
  out[5] = MR * Fp * a;                        // ydrop
  out[6] = MR * kOutDim * 4;                   // feat
  return GIC_OK;
}

int gic_disc_bwd_ws_bytes(const gic_disc_dims* dims, uint64_t* out) {
  DCtx c;
  GIC_PROPAGATE(make_ctx(dims, nullptr, nullptr, c));
  GIC_CHECK_ARG(out, "disc_bwd_ws_bytes: null out");
  const uint64_t a = dtype_size(c.dt), MR = c.rowsBR, Fp = c.Fp;
  out[0] = MR * kOutPad * a;                   // dfeat
  out[1] = MR * Fp * a;                        // dh
  out[2] = MR * Fp * 4;                        // dydrop
  out[3] = MR * Fp * 4;                        // dpooled
  out[4] = (uint64_t)c.rowsBL * c.De * a;      // demb
  return GIC_OK;
}

// shadow.emb [De,V]; shadow.hw_w [Fp,Fp] and shadow.f2o_w [104,Fp] zero-padded images of highway / feature2out
int gic_disc_prepare(const gic_disc_dims* dims, const gic_disc_params* P, const gic_disc_shadow* S, void* stream_) {
  DCtx c;
  GIC_PROPAGATE(make_ctx(dims, P, nullptr, c));
  GIC_CHECK_ARG(P && S && P->emb && P->hw_w && P->f2o_w && S->emb && S->hw_w && S->f2o_w, "disc_prepare: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  if ((const void*)S->emb != (const void*)P->emb)
    GIC_PROPAGATE(cast2d(P->emb, DT_F32, c.V, S->emb, c.dt, c.V, c.De, c.V, stream));
  const size_t a = dtype_size(c.dt);
  GIC_PROPAGATE(fill_zero(S->hw_w, (size_t)c.Fp * c.Fp * a, stream));
  GIC_PROPAGATE(cast2d(P->hw_w, DT_F32, c.F, S->hw_w, c.dt, c.Fp, c.F, c.F, stream));
  if (S->hw_w_t) GIC_PROPAGATE(transpose2d(S->hw_w, S->hw_w_t, c.dt, c.Fp, c.Fp, stream));
  GIC_PROPAGATE(fill_zero(S->f2o_w, (size_t)kOutPad * c.Fp * a, stream));
  GIC_PROPAGATE(cast2d(P->f2o_w, DT_F32, c.F, S->f2o_w, c.dt, c.Fp, kOutDim, c.F, stream));
  return GIC_OK;
}

int gic_disc_fwd(const gic_disc_dims* dims, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* st,
                 const void* inp_soft, int64_t ld_inp, const int64_t* inp_ids, int train, const uint8_t* keep_mask,
                 uint64_t seed, float* logits, const gic_step_scalars* dev_scalars, int seed_slot, void* stream) {
  DCtx c;
  GIC_CHECK_ARG(!dev_scalars || (seed_slot >= 0 && seed_slot < GIC_STEP_SEEDS), "disc_fwd: seed_slot out of range");
  const uint64_t* seed_dev = dev_scalars ? &dev_scalars->seed[seed_slot] : nullptr;
  GIC_PROPAGATE(make_ctx(dims, P, nullptr, c));
  GIC_CHECK_ARG(P && S && st && logits, "disc_fwd: null argument");
  GIC_CHECK_ARG((inp_soft != nullptr) != (inp_ids != nullptr), "disc_fwd: pass exactly one of inp_soft / inp_ids");
  GIC_CHECK_ARG(!inp_soft || ld_inp >= c.V, "disc_fwd: ld_inp < V");
  // forward only (eval mode, no backward to follow): state->argmax and state->hpre may be NULL and are then not written
  GIC_CHECK_ARG(st->emb && st->pooled && st->ydrop && st->feat && (!train || (st->keep && st->argmax && st->hpre)), "disc_fwd: null state buffer");
  GIC_CHECK_ARG(P->emb && P->hw_b && P->f2o_b && P->o2l_w && P->o2l_b && S->emb && S->hw_w && S->f2o_w, "disc_fwd: null parameter");
  if (c.dt == DT_F32)
    return disc_fwd_t<float>(c, P, S, st, inp_soft, ld_inp, inp_ids, train, keep_mask, seed, seed_dev, logits, (hipStream_t)stream);
  return disc_fwd_t<bf16_t>(c, P, S, st, inp_soft, ld_inp, inp_ids, train, keep_mask, seed, seed_dev, logits, (hipStream_t)stream);
}

int gic_disc_fwd_redrop(const gic_disc_dims* dims, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* src,
                        const gic_disc_state* dst, int train, const uint8_t* keep_mask, uint64_t seed, float* logits,
                        const gic_step_scalars* dev_scalars, int seed_slot, void* stream) {
  DCtx c;
  GIC_CHECK_ARG(!dev_scalars || (seed_slot >= 0 && seed_slot < GIC_STEP_SEEDS), "disc_fwd_redrop: seed_slot out of range");
  const uint64_t* seed_dev = dev_scalars ? &dev_scalars->seed[seed_slot] : nullptr;
  GIC_PROPAGATE(make_ctx(dims, P, nullptr, c));
  GIC_CHECK_ARG(P && S && src && dst && logits, "disc_fwd_redrop: null argument");
  GIC_CHECK_ARG(src->pooled && src->hpre && dst->ydrop && dst->feat && (!train || dst->keep), "disc_fwd_redrop: null state buffer");
  GIC_CHECK_ARG(P->f2o_b && P->o2l_w && P->o2l_b && S->f2o_w, "disc_fwd_redrop: null parameter");
  if (c.dt == DT_F32)
    return disc_fwd_redrop_t<float>(c, P, S, src, dst, train, keep_mask, seed, seed_dev, logits, (hipStream_t)stream);
  return disc_fwd_redrop_t<bf16_t>(c, P, S, src, dst, train, keep_mask, seed, seed_dev, logits, (hipStream_t)stream);
}

int gic_disc_bwd(const gic_disc_dims* dims, const gic_disc_params* P, const gic_disc_shadow* S, const gic_disc_state* st,
                 const gic_disc_bwd_ws* ws, const void* inp_soft, int64_t ld_inp, const int64_t* inp_ids, int train,
                 const float* d_logits, const gic_disc_grads* G, int accumulate, void* d_inp, int64_t ld_dinp, void* stream) {
  DCtx c;
  GIC_PROPAGATE(make_ctx(dims, P, G, c));
  GIC_CHECK_ARG(P && S && st && ws && d_logits, "disc_bwd: null argument");
  GIC_CHECK_ARG(inp_soft || inp_ids, "disc_bwd: pass inp_soft, inp_ids, or both (mixed batch: ids for the first B/2 captions)");
  GIC_CHECK_ARG(!(inp_soft && inp_ids) || (G && !d_inp && c.B % 2 == 0), "disc_bwd: a mixed batch needs an even B, grads and no d_inp");
  GIC_CHECK_ARG(ws->dfeat && ws->dh && ws->dydrop && ws->dpooled && ws->demb, "disc_bwd: null workspace buffer");
  GIC_CHECK_ARG(!d_inp || (inp_soft && ld_dinp >= c.V), "disc_bwd: d_inp needs a soft input and ld_dinp >= V");
  if (G) {
    GIC_CHECK_ARG(G->emb && G->hw_w && G->hw_b && G->f2o_w && G->f2o_b && G->o2l_w && G->o2l_b, "disc_bwd: null grad buffer");
    for (int k = 0; k < c.cm.nconv; ++k) GIC_CHECK_ARG(G->conv_w[k] && G->conv_b[k], "disc_bwd: null conv %d grad buffer", k);
  }
  if (c.dt == DT_F32)
    return disc_bwd_t<float>(c, P, S, st, ws, inp_soft, ld_inp, inp_ids, train, d_logits, G, accumulate, d_inp, ld_dinp, (hipStream_t)stream);
  return disc_bwd_t<bf16_t>(c, P, S, st, ws, inp_soft, ld_inp, inp_ids, train, d_logits, G, accumulate, d_inp, ld_dinp, (hipStream_t)stream);
}

}  // extern "C"
