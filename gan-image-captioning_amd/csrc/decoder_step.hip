// Fused per-timestep kernels of Decoder.sample (reference src/generator.py:60-76); design notes in decoder_step.h.
//
// MFMA operand conventions used below (16x16 tiles; lr = lane & 15, lg = lane >> 4):
//   A / B fragment of a 32-deep k-step: lane (lr, lg) holds row lr, k = 8*lg .. 8*lg+7 (8 consecutive elements = one
//   16-byte load in bf16, two in f32); bf16: one v_mfma_f32_16x16x32_bf16; f32 (parity mode): eight v_mfma_f32_16x16x4_f32,
//   sub-step s contracting k = 8*lg + s of every lane group (exact f32 products, f32 accumulation).
//   C tile: lane (lr, lg) holds column lr (B-operand row), rows 4*lg .. 4*lg+3 (A-operand rows) in its 4 registers.
#include "decoder_step.h"

#include <stdlib.h>

#include "kernels.h"

namespace gic {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;
typedef const __attribute__((address_space(1))) f32x4* gptr_f4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <typename TA> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

template <typename TA> __device__ __forceinline__ Frag<TA> zero_frag();
template <> __device__ __forceinline__ Frag<bf16_t> zero_frag<bf16_t>() {
  Frag<bf16_t> f;
  const u32x4 z = {0u, 0u, 0u, 0u};
  f.v = __builtin_bit_cast(bf16x8, z);
  return f;
}
template <> __device__ __forceinline__ Frag<float> zero_frag<float>() {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = 0.f;
  return f;
}

// 8 consecutive elements from global memory (16-byte aligned)
template <typename TA> __device__ __forceinline__ Frag<TA> load_frag(const TA* p);
template <> __device__ __forceinline__ Frag<bf16_t> load_frag<bf16_t>(const bf16_t* p) {
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8, *(gptr_u4)p);
  return f;
}
template <> __device__ __forceinline__ Frag<float> load_frag<float>(const float* p) {
  Frag<float> f;
  const f32x4 a = *(gptr_f4)p, b = *(gptr_f4)(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
  return f;
}
// the same from shared memory
template <typename TA> __device__ __forceinline__ Frag<TA> lds_frag(const unsigned char* p);
template <> __device__ __forceinline__ Frag<bf16_t> lds_frag<bf16_t>(const unsigned char* p) {
  Frag<bf16_t> f;
  f.v = *(const bf16x8*)p;
  return f;
}
template <> __device__ __forceinline__ Frag<float> lds_frag<float>(const unsigned char* p) {
  Frag<float> f;
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 16);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}
// 8 consecutive f32 master values (an embedding row) as a compute-dtype fragment
template <typename TA> __device__ __forceinline__ Frag<TA> frag_from_f32(const float* p);
template <> __device__ __forceinline__ Frag<float> frag_from_f32<float>(const float* p) { return load_frag<float>(p); }
template <> __device__ __forceinline__ Frag<bf16_t> frag_from_f32<bf16_t>(const float* p) {
  const Frag<float> s = load_frag<float>(p);
  Frag<bf16_t> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = (bf16_t)s.v[i];
  return f;
}

template <typename TA> __device__ __forceinline__ void mma(f32x4& acc, const Frag<TA>& a, const Frag<TA>& b);
template <> __device__ __forceinline__ void mma<bf16_t>(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
}

// (value, index) argmax across the 64 lanes: larger value wins, equal values -> smaller index (first maximal index)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// lstm_step: one LSTM layer, one time step, for 64 batch rows x 4 hidden units per block.
//   gates[b, g*H + j] = sum_k [x_t | h_{t-1}][b, k] * Wcat[g*H + j, k] + bsum     (generator.py:61, nn.LSTM cell)
// The 8 waves split K (each owns every 8th group of KPI 32-deep k-steps): operands go straight from L2 to registers (no
// operand is shared between waves, so LDS staging would only add a hop), all of a wave's loads are in flight together,
// and the K partials are summed through LDS by the 256 threads that then apply the cell nonlinearities.
template <typename TA, int KPI>
__global__ __launch_bounds__(512) void lstm_step_kernel(const LstmStepArgs a) {
  __shared__ float red[8][kStepRows][17];
  __shared__ int ids_s[kStepRows];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int j0 = blockIdx.x * kUnitsPerBlock;
  const int b0 = blockIdx.y * kStepRows;

  if (a.gather) {
    // next-input token of each row: the forced trajectory, else the argmax over the previous step's per-tile partials
    // (generator.py:73: first maximal index; tiles are in vocabulary order, so ties go to the smaller index)
    for (int r = w * 8; r < w * 8 + 8; ++r) {
      const int b = b0 + r;
      int id = 0;
      if (b < a.B) {
        const bool forced = a.force_ids && (!a.force_len || a.tprev < a.force_len[b]);
        if (forced) {
          id = (int)a.force_ids[(long)b * a.force_stride + a.tprev];
        } else if (a.part_m) {
          float bm = -INFINITY;
          int bi = 0x7fffffff;
          for (int j = lane; j < a.nblk; j += 64) {
            const float m = a.part_m[(long)b * a.nblk + j];
            const int i = a.part_i[(long)b * a.nblk + j];
            if (m > bm || (m == bm && i < bi)) { bm = m; bi = i; }
          }
          wave_argmax(bm, bi);
          id = bi;
        }
        id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
      }
      if (lane == 0) ids_s[r] = id;
    }
    __syncthreads();
  }

  const TA* xh = (const TA*)a.xh_t;
  const int ju = j0 + (lr & 3);                       // B-operand row lr = gate (lr >> 2), unit (lr & 3)
  const bool jok = ju < a.H;
  const TA* wrow = (const TA*)a.wcat + (long)((lr >> 2) * a.H + (jok ? ju : 0)) * a.ldx;
  const TA* arow[4];
  const float* erow[4];
  bool aok[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int b = b0 + mt * 16 + lr;
    aok[mt] = b < a.B;
    arow[mt] = xh + (long)(aok[mt] ? b : 0) * a.ldx;
    erow[mt] = a.gather ? a.embed + (long)ids_s[mt * 16 + lr] * a.din : nullptr;
  }

  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nks = (int)((a.ldx + 31) >> 5);
  for (int ks0 = w * KPI; ks0 < nks; ks0 += 8 * KPI) {
    Frag<TA> fa[KPI][4], fb[KPI];
#pragma unroll
    for (int i = 0; i < KPI; ++i) {
      const int k = (ks0 + i) * 32 + lg * 8;
      const bool kok = k < a.ldx;                      // ldx % 8 == 0: a fragment is wholly inside or outside
      fb[i] = zero_frag<TA>();
      if (kok && jok) fb[i] = load_frag<TA>(wrow + k);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        fa[i][mt] = zero_frag<TA>();
        if (kok && aok[mt]) {
          if (a.gather && k < a.din) fa[i][mt] = frag_from_f32<TA>(erow[mt] + k);      // din % 8 == 0: never straddles x | h
          else fa[i][mt] = load_frag<TA>(arow[mt] + k);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < KPI; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) mma<TA>(acc[mt], fa[i][mt], fb[i]);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][mt * 16 + lg * 4 + r][lr] = acc[mt][r];
  __syncthreads();

  if (tid < kStepRows * kUnitsPerBlock) {
    const int rb = tid >> 2, u = tid & 3;
    const int b = b0 + rb, j = j0 + u;
    if (b < a.B && j < a.H) {
      float g4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = a.bsum[g * a.H + j];
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) s += red[ww][rb][g * 4 + u];
        g4[g] = s;
      }
      const float i_ = sigmoidf_(g4[0]), f_ = sigmoidf_(g4[1]), g_ = tanhf(g4[2]), o_ = sigmoidf_(g4[3]);
      const long bh = (long)b * a.H + j;
      const float c = f_ * a.c_prev[bh] + i_ * g_;
      const float h = o_ * tanhf(c);
      if (a.gates) {
        float* go = a.gates + (long)b * 4 * a.H + j;
        go[0] = i_; go[a.H] = f_; go[2 * a.H] = g_; go[3 * a.H] = o_;
      }
      a.c_new[bh] = c;
      const TA hv = from_f32<TA>(h);
      ((TA*)a.xh_next)[(long)b * a.ldx + a.din + j] = hv;
      if (a.h_up) ((TA*)a.h_up)[(long)b * a.ld_up + j] = hv;
      if (a.h_out) ((TA*)a.h_out)[(long)b * a.ld_out + j] = hv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// vocab_step: logits^T tile [64 vocabulary entries x 64 batch rows] = W_out[v0:v0+64, :] . h_t^T on MFMA (generator.py:68),
// then y = (o + b_out + gumbel(u)) * T (generator.py:69, 84-96), the tile's softmax partials and e = exp(y - tile max).
// The A operand (weights) is this block's own L2-resident slice, loaded straight into fragments; the B operand (h_t, shared by
// the four vocabulary sub-tiles) is staged once through LDS.  Waves = 4 vocabulary sub-tiles x 2 K halves; the halves trade
// two batch sub-tiles each, so that all eight waves share the transcendental-heavy epilogue.
// In the C tile a lane holds 4 CONSECUTIVE vocabulary entries of one batch row: one Philox4x32 call (or one 16-byte load of
// explicit uniforms) and one vector store of e per lane and batch sub-tile.
template <typename TA, bool FAST>
__global__ __launch_bounds__(512) void vocab_step_kernel(const VocabStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_smem[];
  constexpr int SZ = sizeof(TA);
  const int H = a.H, V = a.V;
  const int hs = H * SZ + 16;                              // LDS row stride of the h tile: 16-B skew against bank conflicts
  unsigned char* sH = vs_smem;                             // [64][hs]
  f32x4* sX = (f32x4*)(vs_smem + kStepRows * hs);          // [8 waves][2 tiles][64 lanes]
  float* red_m = (float*)(sX + 8 * 2 * 64);                // [4][64] each
  float* red_s = red_m + 4 * kStepRows;
  float* red_y = red_s + 4 * kStepRows;
  int* red_i = (int*)(red_y + 4 * kStepRows);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int mt = w >> 1, kh = w & 1;
  const int v0 = blockIdx.x * kVocabTile, b0 = blockIdx.y * kStepRows;

  // ---- this wave's weight fragments: vocabulary row v0 + 16 mt + lr, k-steps kh, kh+2, ... (up to 8 per chunk)
  const int vrow = v0 + mt * 16 + lr;
  const bool vok = vrow < V;
  const TA* wrow = (const TA*)a.wout + (long)(vok ? vrow : 0) * H;
  const int nks = (H + 31) >> 5;
  Frag<TA> fa[8];
  auto load_a = [&](int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = (kh + 2 * (c * 8 + i)) * 32 + lg * 8;
      fa[i] = zero_frag<TA>();
      if (vok && k < H) fa[i] = load_frag<TA>(wrow + k);
    }
  };
  load_a(0);

  // ---- stage h_t [64, H] (rows past B: zero)
  {
    const int cpr = H * SZ / 16;
    const unsigned char* hb = (const unsigned char*)a.h;
    for (int c = tid; c < kStepRows * cpr; c += 512) {
      const int row = c / cpr, cc = c - row * cpr;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (b0 + row < a.B) v = *(gptr_u4)(hb + ((long)(b0 + row) * a.ldh) * SZ + cc * 16);
      *(u32x4*)(sH + row * hs + cc * 16) = v;
    }
  }
  __syncthreads();

  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nhalf = (nks - kh + 1) >> 1;                   // k-steps of this K half
  for (int c = 0; c * 8 < nhalf; ++c) {
    if (c) load_a(c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = (kh + 2 * (c * 8 + i)) * 32 + lg * 8;
      if (c * 8 + i < nhalf) {                             // wave-uniform
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          Frag<TA> fb = zero_frag<TA>();
          if (k < H) fb = lds_frag<TA>(sH + (nt * 16 + lr) * hs + k * SZ);
          mma<TA>(acc[nt], fa[i], fb);
        }
      }
    }
  }

  // ---- K halves: wave (mt, kh) keeps batch sub-tiles 2kh, 2kh+1 and receives the partner's partial sums for them
  // (element-wise selects: a select between accumulator ARRAY elements would become a dynamic index and move them to scratch)
  f32x4 give[2], own[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    give[0][r] = kh ? acc[0][r] : acc[2][r];
    give[1][r] = kh ? acc[1][r] : acc[3][r];
    own[0][r] = kh ? acc[2][r] : acc[0][r];
    own[1][r] = kh ? acc[3][r] : acc[1][r];
  }
  sX[(w * 2 + 0) * 64 + lane] = give[0];
  sX[(w * 2 + 1) * 64 + lane] = give[1];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 p = sX[((w ^ 1) * 2 + i) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) own[i][r] += p[r];
  }

  // ---- epilogue: lane = batch row b0 + 16 nt + lr, vocabulary entries vq .. vq+3
  const int vq = v0 + mt * 16 + lg * 4;
  const bool qok = vq < V;                                 // V % 4 == 0: a quad is wholly inside or outside
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (qok) bias4 = *(gptr_f4)(a.bias + vq);
  const float bia[4] = {bias4[0], bias4[1], bias4[2], bias4[3]};
  const float eps = 1e-10f;                                // generator.py:84
  float y[2][4];
  bool bok[2];
  int brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int nt = 2 * kh + i;
    brow[i] = b0 + nt * 16 + lr;
    bok[i] = brow[i] < a.B;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[i][r] = -INFINITY;
    if (bok[i] && qok) {
      float uu[4] = {0.f, 0.f, 0.f, 0.f};
      if (!a.pretrain) {
        if (a.u) {
          const f32x4 uv = *(gptr_f4)(a.u + (long)brow[i] * V + vq);
          uu[0] = uv[0]; uu[1] = uv[1]; uu[2] = uv[2]; uu[3] = uv[3];
        } else {
          uint32_t r0, r1, r2, r3;
          Philox::gen4(a.seed, a.rng_stream, (uint64_t)brow[i] * (uint64_t)(V >> 2) + (uint64_t)(vq >> 2), r0, r1, r2, r3);
          uu[0] = Philox::u01(r0); uu[1] = Philox::u01(r1); uu[2] = Philox::u01(r2); uu[3] = Philox::u01(r3);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o = own[i][r] + bia[r];
        if (!a.pretrain) {
          const float g = FAST ? -__logf(-__logf(uu[r] + eps) + eps) : -logf(-logf(uu[r] + eps) + eps);
          o = (o + g) * a.temperature;
        }
        y[i][r] = o;
      }
    }
  }
  // tile max and first maximal index per batch row: over the lane's quad, the 4 lane groups, then the 4 vocabulary sub-tiles
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bv = y[i][0];
    int bi = vq;
#pragma unroll
    for (int r = 1; r < 4; ++r)
      if (y[i][r] > bv) { bv = y[i][r]; bi = vq + r; }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lg == 0) { red_y[mt * kStepRows + (2 * kh + i) * 16 + lr] = bv; red_i[mt * kStepRows + (2 * kh + i) * 16 + lr] = bi; }
  }
  __syncthreads();
  float e[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = (2 * kh + i) * 16 + lr;
    const float m = fmaxf(fmaxf(red_y[col], red_y[kStepRows + col]), fmaxf(red_y[2 * kStepRows + col], red_y[3 * kStepRows + col]));
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[i][r] = a.pretrain ? y[i][r] : (FAST ? __expf(y[i][r] - m) : expf(y[i][r] - m));      // exp(-inf) = 0 outside the vocabulary
      s += a.pretrain ? 0.f : e[i][r];
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lg == 0) red_s[mt * kStepRows + col] = s;
    if (a.out && bok[i] && qok) {
      TA o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = from_f32<TA>(e[i][r]);
      TA* dst = (TA*)a.out + (long)brow[i] * a.out_stride + vq;
      if constexpr (sizeof(TA) == 4) *(float4*)dst = *(const float4*)o4;
      else *(uint2*)dst = *(const uint2*)o4;
    }
  }
  __syncthreads();
  if (tid < kStepRows && b0 + tid < a.B) {
    float m = red_y[tid], s = red_s[tid];
    int bi = red_i[tid];
#pragma unroll
    for (int q = 1; q < 4; ++q) {                          // sub-tiles are in vocabulary order: a tie keeps the earlier one
      const float ym = red_y[q * kStepRows + tid];
      if (ym > m) { m = ym; bi = red_i[q * kStepRows + tid]; }
      s += red_s[q * kStepRows + tid];
    }
    const long o = (long)(b0 + tid) * a.nblk + blockIdx.x;
    a.part_m[o] = m;
    a.part_s[o] = s;
    a.part_i[o] = bi;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// sample_finish: one block per (step, caption).  Global max / sum from the tile partials; token id (argmax, or the forced one);
// p = e * exp(tile max - global max) / global sum (generator.py:69: softmax over the whole vocabulary); the embedding row of the
// token into the next step's x slot of XH_0 (operand of the LSTM weight gradient).
template <typename TA>
__global__ __launch_bounds__(256) void sample_finish_kernel(const SampleFinishArgs a) {
  __shared__ float red[16];
  __shared__ float red_v[4];
  __shared__ int red_i[4];
  __shared__ float scale_s[1024];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = blockIdx.x / a.B, b = blockIdx.x % a.B;
  const long po = ((long)t * a.B + b) * a.nblk;
  const float* pm = a.part_m + po;
  float bm = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = tid; j < a.nblk; j += 256) {
    const float m = pm[j];
    const int i = a.part_i[po + j];
    if (m > bm || (m == bm && i < bi)) { bm = m; bi = i; }
  }
  wave_argmax(bm, bi);
  if (lane == 0) { red_v[w] = bm; red_i[w] = bi; }
  __syncthreads();
  bm = red_v[0]; bi = red_i[0];
  for (int q = 1; q < 4; ++q)
    if (red_v[q] > bm || (red_v[q] == bm && red_i[q] < bi)) { bm = red_v[q]; bi = red_i[q]; }
  int id = bi;
  if (a.force_ids && (!a.force_len || t < a.force_len[b])) id = (int)a.force_ids[(long)b * a.L + t];
  id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
  if (tid == 0) a.ids[(long)b * a.L + t] = id;
  if (a.xh0 && t + 1 < a.L) {
    TA* dst = (TA*)a.xh0 + ((long)(t + 1) * a.B + b) * a.ldx0;
    for (int e = tid; e < a.E; e += 256) dst[e] = from_f32<TA>(a.embed[(long)id * a.E + e]);
  }
  if (!a.out || a.pretrain) return;
  float s = 0.f;
  for (int j = tid; j < a.nblk; j += 256) s += a.part_s[po + j] * expf(pm[j] - bm);
  s = block_sum(s, red);
  const float inv = 1.f / s;
  for (int j = tid; j < a.nblk; j += 256) scale_s[j] = expf(pm[j] - bm) * inv;
  __syncthreads();
  TA* row = (TA*)a.out + ((long)b * a.L + t) * a.V;
  for (int q = tid; q < (a.V >> 2); q += 256) {
    const float sc = scale_s[(4 * q) / kVocabTile];
    TA v4[4];
    if constexpr (sizeof(TA) == 4) *(float4*)v4 = *(const float4*)(row + 4 * q);
    else *(uint2*)v4 = *(const uint2*)(row + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = from_f32<TA>(to_f32<TA>(v4[r]) * sc);
    if constexpr (sizeof(TA) == 4) *(float4*)(row + 4 * q) = *(const float4*)v4;
    else *(uint2*)(row + 4 * q) = *(const uint2*)v4;
  }
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return GIC_OK;
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_last_error("vocab_step: cannot reserve %zu bytes of LDS", bytes);
    return GIC_ERR_LAUNCH;
  }
  return GIC_OK;
}

size_t vocab_lds_bytes(int dtype, int H) {
  return (size_t)kStepRows * ((size_t)H * dtype_size(dtype) + 16) + 8 * 2 * 64 * 16 + 4 * 4 * kStepRows * 4;
}

}  // namespace

bool decoder_step_supported(int dtype, int V, int E, int H, int NL) {
  static const bool off = getenv("GIC_NO_FUSED_ROLLOUT") != nullptr;
  if (off) return false;
  if (V < 4 || V % 4 || E % 8 || H % 8 || NL < 1) return false;
  if ((V + kVocabTile - 1) / kVocabTile > 1024) return false;          // sample_finish's scale table
  return vocab_lds_bytes(dtype, H) <= 160 * 1024 - 1024;
}

size_t decoder_step_part_floats(int B, int L, int V) {
  return (size_t)3 * L * B * ((V + kVocabTile - 1) / kVocabTile);
}

int lstm_step(const LstmStepArgs& a, int dtype, hipStream_t stream) {
  GIC_CHECK_ARG(a.xh_t && a.xh_next && a.wcat && a.bsum && a.c_prev && a.c_new, "lstm_step: null buffer");
  GIC_CHECK_ARG(a.B > 0 && a.H > 0 && a.din > 0 && a.ldx == (long)a.din + a.H && a.ldx % 8 == 0 && a.din % 8 == 0, "lstm_step: bad dims");
  GIC_CHECK_ARG(!a.gather || (a.embed && a.V > 0 && (a.part_m || a.force_ids) && (!a.part_m || (a.part_i && a.nblk > 0))), "lstm_step: bad gather arguments");
  const dim3 grid((unsigned)cdiv(a.H, kUnitsPerBlock), (unsigned)cdiv(a.B, kStepRows));
  if (dtype == DT_F32) hipLaunchKernelGGL((lstm_step_kernel<float, 2>), grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((lstm_step_kernel<bf16_t, 4>), grid, dim3(512), 0, stream, a);
  GIC_CHECK_LAUNCH("lstm_step");
  return GIC_OK;
}

int vocab_step(const VocabStepArgs& a, int dtype, hipStream_t stream) {
  GIC_CHECK_ARG(a.h && a.wout && a.bias && a.part_m && a.part_s && a.part_i, "vocab_step: null buffer");
  GIC_CHECK_ARG(a.B > 0 && a.V >= 4 && a.V % 4 == 0 && a.H % 8 == 0 && a.ldh % 8 == 0, "vocab_step: bad dims");
  GIC_CHECK_ARG(a.nblk == cdiv(a.V, kVocabTile), "vocab_step: nblk must be ceil(V / %d)", kVocabTile);
  GIC_CHECK_ARG(!a.out || a.out_stride % 4 == 0, "vocab_step: out row stride must be a multiple of 4");
  const size_t lds = vocab_lds_bytes(dtype, a.H);
  const dim3 grid((unsigned)a.nblk, (unsigned)cdiv(a.B, kStepRows));
  if (dtype == DT_F32) {
    GIC_PROPAGATE(allow_lds(vocab_step_kernel<float, false>, lds));
    hipLaunchKernelGGL((vocab_step_kernel<float, false>), grid, dim3(512), lds, stream, a);
  } else {
    GIC_PROPAGATE(allow_lds(vocab_step_kernel<bf16_t, true>, lds));
    hipLaunchKernelGGL((vocab_step_kernel<bf16_t, true>), grid, dim3(512), lds, stream, a);
  }
  GIC_CHECK_LAUNCH("vocab_step");
  return GIC_OK;
}

int sample_finish(const SampleFinishArgs& a, int dtype, hipStream_t stream) {
  GIC_CHECK_ARG(a.part_m && a.part_s && a.part_i && a.ids && a.nblk > 0 && a.nblk <= 1024, "sample_finish: bad partials");
  GIC_CHECK_ARG(a.B > 0 && a.L > 0 && a.V % 4 == 0, "sample_finish: bad dims");
  GIC_CHECK_ARG(!a.xh0 || a.embed, "sample_finish: the x rows need the embedding table");
  const dim3 grid((unsigned)((long)a.B * a.L));
  if (dtype == DT_F32) hipLaunchKernelGGL((sample_finish_kernel<float>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((sample_finish_kernel<bf16_t>), grid, dim3(256), 0, stream, a);
  GIC_CHECK_LAUNCH("sample_finish");
  return GIC_OK;
}

}  // namespace gic
