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

#ifdef GIC_STAMPS
}  // anon
__device__ unsigned long long g_dstamp[2][2][8][2];      // [lstm_step | vocab_step][first | last block][phase][shader clock, 100 MHz real time]
namespace {
#define DSTAMP(kn, i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) { \
  unsigned long long* p_ = g_dstamp[kn][blockIdx.x == 0 ? 0 : 1][i]; p_[0] = __builtin_amdgcn_s_memtime(); p_[1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define DSTAMP(kn, i) do {} while (0)
#endif

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

// (row_key / row_key_index: common.h)

// (value, index) argmax across the 64 lanes: larger value wins, equal values -> smaller index (first maximal index)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// 16 bytes of the compute-dtype image of an f32 master row (embedding gather) from its raw 16-byte loads: 8 values in bf16 (two
// loads), 4 in f32 (one).  Loading and converting are separate: a thread has every load of a pass in flight before it touches one.
template <typename TA> __device__ __forceinline__ u32x4 pack_f32_chunk(const f32x4& lo, const f32x4& hi);
template <> __device__ __forceinline__ u32x4 pack_f32_chunk<float>(const f32x4& lo, const f32x4& hi) { return __builtin_bit_cast(u32x4, lo); }
template <> __device__ __forceinline__ u32x4 pack_f32_chunk<bf16_t>(const f32x4& lo, const f32x4& hi) {
  bf16x8 v;
  v[0] = (bf16_t)lo[0]; v[1] = (bf16_t)lo[1]; v[2] = (bf16_t)lo[2]; v[3] = (bf16_t)lo[3];
  v[4] = (bf16_t)hi[0]; v[5] = (bf16_t)hi[1]; v[6] = (bf16_t)hi[2]; v[7] = (bf16_t)hi[3];
  return __builtin_bit_cast(u32x4, v);
}

// ------------------------------------------------------------------------------------------------------------------
// lstm_step: one LSTM layer, one time step, for 64 batch rows x 4 hidden units per block.
//   gates[b, g*H + j] = sum_k [x_t | h_{t-1}][b, k] * Wcat[g*H + j, k] + bsum     (generator.py:61, nn.LSTM cell)
// The activation tile [64, K] reaches LDS in K chunks of KC by COALESCED 16-byte loads (consecutive lanes = consecutive chunks
// of one row; the x columns of a gathering step come from the f32 embedding rows, converted on the way) and MFMA fragments
// are read from there: loading fragments straight from global memory touches 16 rows per wave instruction and is bound by
// the texture addresser (measured 3x slower).  The 16 weight rows of the block are few enough to go straight to fragments.
// The 8 waves split the k-steps of a chunk; their partial sums meet in LDS (the staging area, reused) and 256 threads apply the
// cell nonlinearities.
// (device body: workgroup (bx, by) of the launch grid)
template <typename TA>
__device__ __forceinline__ void lstm_step_body(const LstmStepArgs& a, const int KC, const int bx, const int by, unsigned char* ls_smem, int* ids_s) {
  constexpr int SZ = sizeof(TA), VE = 16 / SZ;
  const int hs = KC * SZ + 16;                              // LDS row stride: 16-byte skew against bank conflicts
  unsigned char* sA = ls_smem;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int j0 = bx * kUnitsPerBlock;
  const int b0 = by * kStepRows;
  DSTAMP(0, 0);

  if (a.gather) {
    // next-input token of each row: the forced trajectory, else the first maximal index of the previous step's logits
    // (generator.py:73), left by vocab_step as a 64-bit atomicMax key per row
    if (tid < kStepRows) {
      const int b = b0 + tid;
      int id = 0;
      if (b < a.B) {
        const bool forced = a.force_ids && (!a.force_len || a.tprev < a.force_len[b]);
        if (forced) id = (int)a.force_ids[(long)b * a.force_stride + a.tprev];
        else if (a.rowkey) id = row_key_index(a.rowkey[b]);
        id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
      }
      ids_s[tid] = id;
    }
    __syncthreads();
  }

  DSTAMP(0, 1);
  const TA* xh = (const TA*)a.xh_t;
  const int ju = j0 + (lr & 3);                       // B-operand row lr = gate (lr >> 2), unit (lr & 3)
  const bool jok = ju < a.H;
  const TA* wrow = (const TA*)a.wcat + (long)((lr >> 2) * a.H + (jok ? ju : 0)) * a.ldx;

  // operands of the pointwise stage (threads 0..255: row tid >> 2, unit tid & 3): requested now, consumed after the products
  const int pb = b0 + (tid >> 2), pj = j0 + (tid & 3);
  const bool pok = tid < kStepRows * kUnitsPerBlock && pb < a.B && pj < a.H;
  float pbias[4] = {0.f, 0.f, 0.f, 0.f}, pc = 0.f;
  if (pok) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pbias[g] = a.bsum[g * a.H + pj];
    pc = a.c_prev[(long)pb * a.H + pj];
  }

  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ldx = (int)a.ldx;
  const int gw = a.gw ? a.gw : a.din;
  // Every block of the launch stages the SAME activation rows: walked in the same order they would all ask one L2 channel for the
  // same lines at the same moment (measured: 7.4 us of a 10.5 us block for 192 KB).  Blocks sharing an XCD (blockIdx % 8) start at
  // different rows, so that at any instant their requests fall on different channels.
  const int rot = ((bx >> 3) * 4 + (bx & 7)) & (kStepRows - 1);
  // One K chunk.  Written as a lambda that the common single-chunk / single-pass shape calls in STRAIGHT-LINE code: inside a loop the
  // compiler's wait-count bookkeeping merges the back edge's pending loads into the loop head and waits for everything outstanding
  // (the bias / cell-state loads, then the weight fragments) before it issues the staging loads -- two extra serial round trips.
  auto chunk = [&](const int kc0, const int kc) {
    // ---- this wave's weight fragments of the chunk (k-steps w, w+8, ...: at most 4 for KC <= 1024), straight from L2; k-steps past
    // the chunk re-read its last one (their activation fragments are zero).  Issued from inside the first staging pass.
    Frag<TA> fb[4];
    auto load_weights = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = load_frag<TA>(wrow + kc0 + min((w + 8 * i) * 32 + lg * 8, kc - 8));
    };
    // ---- stage the activation chunk [64, kc]: the leading `xc` columns of a gathering step come from the f32 embedding rows of the
    // rows' tokens, the rest from [x_t | h_{t-1}].  Per pass a thread has 8 + 8 pieces in flight: UNCONDITIONAL loads from clamped
    // addresses (a predicated or branch-wrapped load makes the compiler wait for each piece in turn: measured 16 serial round trips,
    // 15k cycles), piece coordinates by increments (one integer division per thread and range), then conversions and LDS writes.
    // Rows past B repeat row B-1 (their results are never stored).
    if (!(a.dbg & 2)) {
      const int xc = a.gather ? max(0, min(kc, gw - kc0)) : 0;      // gw % 8 == 0: a 16-byte piece never straddles x | rest
      const int rlast = a.B - 1 - b0;                                // last valid tile row (>= 0: the grid covers B)
      const int cprx = xc / VE, totx = kStepRows * cprx;             // embedding columns [0, xc)
      const int cprh = (kc - xc) / VE, toth = kStepRows * cprh;      // activation columns [xc, kc)
      int rowx = 0, ccx = 0, drowx = 0, dccx = 0, rowh = 0, cch = 0, drowh = 0, dcch = 0;
      if (totx) { rowx = tid / cprx; ccx = tid % cprx; drowx = 512 / cprx; dccx = 512 % cprx; }
      if (toth) { rowh = tid / cprh; cch = tid % cprh; drowh = 512 / cprh; dcch = 512 % cprh; }
      auto pass = [&](const int c0) {
        const bool dox = c0 < totx, doh = c0 < toth;                 // block-uniform
        u32x4 lo[8], hi[8], v[8];
        int prx[8], pcx[8], prh[8], pch[8];
        if (dox) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            prx[i] = rowx; pcx[i] = ccx;
            const int r = min((rowx + rot) & (kStepRows - 1), rlast);
            const float* src = a.embed + (long)ids_s[r] * gw + kc0 + ccx * VE;
            lo[i] = *(gptr_u4)src;
            if (SZ == 2) hi[i] = *(gptr_u4)(src + 4);
            rowx += drowx; ccx += dccx;
            if (ccx >= cprx) { ccx -= cprx; ++rowx; }
            if (rowx >= kStepRows) rowx = kStepRows - 1;             // pieces past the end re-read the last row (not written)
          }
        }
        if (doh) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            prh[i] = rowh; pch[i] = cch;
            const int r = min((rowh + rot) & (kStepRows - 1), rlast);
            v[i] = *(gptr_u4)(xh + (long)(b0 + r) * a.ldx + kc0 + xc + cch * VE);
            rowh += drowh; cch += dcch;
            if (cch >= cprh) { cch -= cprh; ++rowh; }
            if (rowh >= kStepRows) rowh = kStepRows - 1;
          }
        }
        if (c0 == 0) load_weights();                                 // behind the staging loads: nothing waits for them but the MFMAs
        if (dox) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            u32x4 x = lo[i];
            if (SZ == 2) x = pack_f32_chunk<TA>(__builtin_bit_cast(f32x4, lo[i]), __builtin_bit_cast(f32x4, hi[i]));
            if (c0 + i * 512 + tid < totx) *(u32x4*)(sA + ((prx[i] + rot) & (kStepRows - 1)) * hs + pcx[i] * 16) = x;
          }
        }
        if (doh) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (c0 + i * 512 + tid < toth) *(u32x4*)(sA + ((prh[i] + rot) & (kStepRows - 1)) * hs + xc * SZ + pch[i] * 16) = v[i];
        }
      };
      if (totx <= 8 * 512 && toth <= 8 * 512) pass(0);
      else for (int c0 = 0; c0 < totx || c0 < toth; c0 += 8 * 512) pass(c0);
    } else {
      load_weights();
    }
    DSTAMP(0, 2);
    __syncthreads();
    DSTAMP(0, 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kl = (w + 8 * i) * 32 + lg * 8;
      if ((w + 8 * i) * 32 < kc) {                       // wave-uniform
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          Frag<TA> fa = zero_frag<TA>();
          if (kl < kc) fa = lds_frag<TA>(sA + (mt * 16 + lr) * hs + kl * SZ);
          mma<TA>(acc[mt], fa, fb[i]);
        }
      }
    }
  };
  if (ldx <= KC) chunk(0, ldx);
  else
    for (int kc0 = 0; kc0 < ldx; kc0 += KC) {
      if (kc0) __syncthreads();                          // every wave has read the previous chunk
      chunk(kc0, min(KC, ldx - kc0));
    }
  DSTAMP(0, 4);
  __syncthreads();                                       // the staging area becomes the reduction buffer
  float (*red)[kStepRows][17] = (float (*)[kStepRows][17])ls_smem;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][mt * 16 + lg * 4 + r][lr] = acc[mt][r];
  __syncthreads();
  DSTAMP(0, 5);

  if (pok) {
    const int rb = tid >> 2, u = tid & 3;
    const int b = pb, j = pj;
    float g4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = pbias[g];
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) s += red[ww][rb][g * 4 + u];
      g4[g] = s;
    }
    const float i_ = sigmoidf_(g4[0]), f_ = sigmoidf_(g4[1]), g_ = tanhf(g4[2]), o_ = sigmoidf_(g4[3]);
    const long bh = (long)b * a.H + j;
    const float c = f_ * pc + i_ * g_;
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
  DSTAMP(0, 6);
}

template <typename TA>
__global__ __launch_bounds__(512) void lstm_step_kernel(const LstmStepArgs a, const int KC) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ls_smem[];
  __shared__ int ids_s[kStepRows];
  lstm_step_body<TA>(a, KC, blockIdx.x, blockIdx.y, ls_smem, ids_s);
}

// ------------------------------------------------------------------------------------------------------------------
// vocab_step: logits^T tile [64 vocabulary entries x 64 batch rows] = W_out[v0:v0+64, :] . h_t^T on MFMA (generator.py:68),
// then y = (o + b_out + gumbel(u)) * T (generator.py:69, 84-96), the tile's softmax partials and e = exp(y - tile max).
// Both operand tiles (this block's own L2-resident weight slice and h_t) reach LDS in K chunks of KC by coalesced 16-byte loads;
// waves = 4 vocabulary sub-tiles x 2 K halves of a chunk; the halves trade two batch sub-tiles each, so that all eight waves
// share the transcendental-heavy epilogue.  In the C tile a lane holds 4 CONSECUTIVE vocabulary entries of one batch row: one
// Philox4x32 call (or one 16-byte load of explicit uniforms) and one vector store of e per lane and batch sub-tile.
template <typename TA, bool FAST>
__device__ __forceinline__ void vocab_step_body(const VocabStepArgs& a, const int KC, const int bx, const int by, unsigned char* vs_smem) {
  constexpr int SZ = sizeof(TA), VE = 16 / SZ;
  const int H = a.H, V = a.V;
  const int hs = KC * SZ + 16;                             // LDS row stride of both tiles: 16-B skew against bank conflicts
  unsigned char* sW = vs_smem;                             // [64 vocabulary rows][hs]
  unsigned char* sH = vs_smem + kStepRows * hs;            // [64 batch rows][hs]
  // after the products the staging area is reused: exchange buffer + reduction scratch
  f32x4* sX = (f32x4*)vs_smem;                             // [8 waves][2 tiles][64 lanes]
  float* red_m = (float*)(sX + 8 * 2 * 64);                // [4][64] each
  float* red_s = red_m + 4 * kStepRows;
  float* red_y = red_s + 4 * kStepRows;
  int* red_i = (int*)(red_y + 4 * kStepRows);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int mt = w >> 1, kh = w & 1;
  const int v0 = bx * kVocabTile, b0 = by * kStepRows;
  DSTAMP(1, 0);

  // ---- epilogue operands that depend on nothing computed here: requested now
  const int vq = v0 + mt * 16 + lg * 4;
  const bool qok = vq < V;                                 // V % 4 == 0: a quad is wholly inside or outside
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (qok) bias4 = *(gptr_f4)(a.bias + vq);
  f32x4 u4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (a.u && !a.pretrain && qok) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int b = b0 + (2 * kh + i) * 16 + lr;
      if (b < a.B) u4[i] = *(gptr_f4)(a.u + (long)b * V + vq);
    }
  }

  f32x4 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int rot = ((bx >> 3) * 4 + (bx & 7)) & (kStepRows - 1);     // h_t is every block's: see lstm_step
  const TA* Wg = (const TA*)a.wout;
  const TA* Hg = (const TA*)a.h;
  for (int kc0 = 0; kc0 < H; kc0 += KC) {
    const int kc = min(KC, H - kc0);
    if (kc0) __syncthreads();
    // ---- stage W_out[v0:v0+64, kc0:kc0+kc] and h_t[b0:b0+64, kc0:kc0+kc]: 8 + 8 pieces per thread in flight per pass, unconditional
    // loads from clamped addresses and piece coordinates by increments (see lstm_step).  Vocabulary rows past V and batch rows past
    // B repeat the last valid one (their results are masked / never stored); a K tail (columns kc .. kc32) is staged as ZEROS.
    const int kc32 = (kc + 31) & ~31;
    const int cpr = kc32 / VE;
    const int total = kStepRows * cpr;
    if (!(a.dbg & 4)) {
      const int drow = 512 / cpr, dcc = 512 % cpr;
      const int wlast = V - 1 - v0, hlast = a.B - 1 - b0, cclast = kc / VE - 1;
      int row = tid / cpr, cc = tid % cpr;
      for (int c0 = 0; c0 < total; c0 += 8 * 512) {
        u32x4 vw[8], vh[8];
        int prow[8], pcc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          prow[i] = row; pcc[i] = cc;
          const int ccl = min(cc, cclast);
          const int hrow = (row + rot) & (kStepRows - 1);
          vw[i] = *(gptr_u4)(Wg + (long)(v0 + min(row, wlast)) * H + kc0 + ccl * VE);
          vh[i] = *(gptr_u4)(Hg + (long)(b0 + min(hrow, hlast)) * a.ldh + kc0 + ccl * VE);
          row += drow; cc += dcc;
          if (cc >= cpr) { cc -= cpr; ++row; }
          if (row >= kStepRows) row = kStepRows - 1;           // pieces past the end re-read the last row (not written)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool tail = pcc[i] > cclast;
          if (tail) vw[i] = vh[i] = (u32x4){0u, 0u, 0u, 0u};
          if (c0 + i * 512 + tid < total) {
            *(u32x4*)(sW + prow[i] * hs + pcc[i] * 16) = vw[i];
            *(u32x4*)(sH + ((prow[i] + rot) & (kStepRows - 1)) * hs + pcc[i] * 16) = vh[i];
          }
        }
      }
    }
    DSTAMP(1, 1);
    __syncthreads();
    DSTAMP(1, 2);
    // ---- products: the chunk's k-steps split in two contiguous halves
    const int nks = (kc + 31) >> 5;
    const int half0 = (nks + 1) >> 1;
    const int ks_lo = kh ? half0 : 0, ks_hi = kh ? nks : half0;
    // (the K tail is zero in LDS: unconditional fragment reads, the next k-step's five in flight under this one's MFMAs)
    if (!(a.dbg & 2) && ks_lo < ks_hi) {
      const unsigned char* pa = sW + (mt * 16 + lr) * hs + lg * 8 * SZ;
      const unsigned char* pb = sH + lr * hs + lg * 8 * SZ;
      Frag<TA> fa = lds_frag<TA>(pa + ks_lo * 32 * SZ), fb[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fb[nt] = lds_frag<TA>(pb + nt * 16 * hs + ks_lo * 32 * SZ);
      for (int ks = ks_lo; ks < ks_hi; ++ks) {
        const int kn = (ks + 1 < ks_hi ? ks + 1 : ks) * 32 * SZ;      // the last iteration re-reads its own step (discarded)
        const Frag<TA> fa_n = lds_frag<TA>(pa + kn);
        Frag<TA> fb_n[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb_n[nt] = lds_frag<TA>(pb + nt * 16 * hs + kn);
        __builtin_amdgcn_sched_barrier(0);                             // (the scheduler would sink the reads to their uses)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) mma<TA>(acc[nt], fa, fb[nt]);
        __builtin_amdgcn_sched_barrier(0);
        fa = fa_n;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb[nt] = fb_n[nt];
      }
    }
  }
  DSTAMP(1, 3);
  __syncthreads();                                         // staging area -> exchange / reduction scratch

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

  DSTAMP(1, 4);
  // ---- epilogue: lane = batch row b0 + 16 nt + lr, vocabulary entries vq .. vq+3
  const float bia[4] = {bias4[0], bias4[1], bias4[2], bias4[3]};
  const float eps = 1e-10f;                                // generator.py:84
  const float temperature = a.t_dev ? *a.t_dev : a.temperature;      // (scalar loads; device-resident in the replayed step graph)
  const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
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
          uu[0] = u4[i][0]; uu[1] = u4[i][1]; uu[2] = u4[i][2]; uu[3] = u4[i][3];
        } else {
          uint32_t r0, r1, r2, r3;
          Philox::gen4(seed, a.rng_stream, (uint64_t)brow[i] * (uint64_t)(V >> 2) + (uint64_t)(vq >> 2), r0, r1, r2, r3);
          uu[0] = Philox::u01(r0); uu[1] = Philox::u01(r1); uu[2] = Philox::u01(r2); uu[3] = Philox::u01(r3);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o = own[i][r] + bia[r];
        if (!a.pretrain) {
          const float g = FAST ? -__logf(-__logf(uu[r] + eps) + eps) : -logf(-logf(uu[r] + eps) + eps);
          o = (o + g) * temperature;
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
  DSTAMP(1, 5);
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
  DSTAMP(1, 6);
  if (tid < kStepRows && b0 + tid < a.B) {
    float m = red_y[tid], s = red_s[tid];
    int bi = red_i[tid];
#pragma unroll
    for (int q = 1; q < 4; ++q) {                          // sub-tiles are in vocabulary order: a tie keeps the earlier one
      const float ym = red_y[q * kStepRows + tid];
      if (ym > m) { m = ym; bi = red_i[q * kStepRows + tid]; }
      s += red_s[q * kStepRows + tid];
    }
    const long o = (long)(b0 + tid) * a.nblk + bx;
    a.part_m[o] = m;
    a.part_s[o] = s;
    atomicMax(a.rowkey + b0 + tid, row_key(m, bi));
  }
  DSTAMP(1, 7);
}

template <typename TA, bool FAST>
__global__ __launch_bounds__(512) void vocab_step_kernel(const VocabStepArgs a, const int KC) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_smem[];
  vocab_step_body<TA, FAST>(a, KC, blockIdx.x, blockIdx.y, vs_smem);
}

// ------------------------------------------------------------------------------------------------------------------
// sample_finish: one block per (step, caption).  Global max / sum from the tile partials; token id (argmax, or the forced one);
// p = e * exp(tile max - global max) / global sum (generator.py:69: softmax over the whole vocabulary); the embedding row of the
// token into the next step's x slot of XH_0 (operand of the LSTM weight gradient).
template <typename TA>
__global__ __launch_bounds__(256) void sample_finish_kernel(const SampleFinishArgs a) {
  __shared__ float red[16];
  __shared__ float scale_s[1024];
  const int tid = threadIdx.x;
  const int t = blockIdx.x / a.B, b = blockIdx.x % a.B;
  const long po = ((long)t * a.B + b) * a.nblk;
  const float* pm = a.part_m + po;
  int id = row_key_index(a.rowkey[(long)t * a.B + b]);
  if (a.force_ids && (!a.force_len || t < a.force_len[b])) id = (int)a.force_ids[(long)b * a.L + t];
  id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
  if (tid == 0) a.ids[(long)b * a.L + t] = (int64_t)id;
  if (a.xh0 && t + 1 < a.L) {
    TA* dst = (TA*)a.xh0 + ((long)(t + 1) * a.B + b) * a.ldx0;
    for (int e = tid; e < a.E; e += 256) dst[e] = from_f32<TA>(a.embed[(long)id * a.E + e]);
  }
  if (!a.out || a.pretrain) return;
  float bm = -INFINITY;
  for (int j = tid; j < a.nblk; j += 256) bm = fmaxf(bm, pm[j]);
  bm = block_max(bm, red);
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

// ------------------------------------------------------------------------------------------------------------------
// lstm_bwd_step: see decoder_step.h.  C[b, j] = sum over up to two K segments of A_seg[b, k] * W_seg[j, k].
template <typename TA, int KPI>
__global__ __launch_bounds__(512) void lstm_bwd_step_kernel(const LstmBwdStepArgs a) {
  __shared__ float red[8][16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int K = 4 * a.H;
  const bool bok = b0 + lr < a.B, jok = j0 + lr < a.H;
  // operands of the pointwise stage (threads 0..255: row tid >> 4, unit tid & 15): requested now, consumed after the products
  const int pb = b0 + (tid >> 4), pj = j0 + (tid & 15);
  const bool pok = tid < 256 && pb < a.B && pj < a.H;
  const long pbh = (long)(pok ? pb : 0) * a.H + (pok ? pj : 0);
  float p_dh = 0.f, p_g[4] = {0.f, 0.f, 0.f, 0.f}, p_cc = 0.f, p_cp = 0.f, p_dc = 0.f;
  if (pok) {
    if (a.dh_above) p_dh = a.dh_above[(long)pb * a.ld_above + pj];
    if (a.dh_extra) {
      p_dh += a.dh_extra[pbh];
      if (a.zero_extra) const_cast<float*>(a.dh_extra)[pbh] = 0.f;      // (each element has exactly one reader: this thread)
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) p_g[g] = a.gates[(long)pb * K + g * a.H + pj];
    p_cc = a.c_cur[pbh]; p_cp = a.c_prev[pbh]; p_dc = a.dc_state[pbh];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nks = (K + 31) >> 5;
#pragma unroll
  for (int seg = 0; seg < 2; ++seg) {
    const TA* A = (const TA*)(seg ? a.dg_up : a.dg_next);
    const TA* W = (const TA*)(seg ? a.w_up : a.w_rec);
    if (!A) continue;                                    // block-uniform
    const TA* arow = A + (long)(bok ? b0 + lr : 0) * K;
    const TA* wrow = W + (long)(jok ? j0 + lr : 0) * K;
    for (int ks0 = w * KPI; ks0 < nks; ks0 += 8 * KPI) {
      Frag<TA> fa[KPI], fb[KPI];
#pragma unroll
      for (int i = 0; i < KPI; ++i) {        // unconditional loads from clamped addresses: all 2 KPI of them in flight
        const int k = min((ks0 + i) * 32 + lg * 8, K - 8);
        fa[i] = load_frag<TA>(arow + k);
        fb[i] = load_frag<TA>(wrow + k);
      }
#pragma unroll
      for (int i = 0; i < KPI; ++i) {
        if ((ks0 + i) * 32 + lg * 8 >= K) fa[i] = zero_frag<TA>();       // k-steps past K (rows / units past the end are never stored)
        mma<TA>(acc, fa[i], fb[i]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][lg * 4 + r][lr] = acc[r];
  __syncthreads();
  if (pok) {
    const int rb = tid >> 4, u = tid & 15;
    const int j = pj;
    float dh = p_dh;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) dh += red[ww][rb][u];
    const float i_ = p_g[0], f_ = p_g[1], g_ = p_g[2], o_ = p_g[3];
    const float tc = tanhf(p_cc);
    const float dc = p_dc + dh * o_ * (1.f - tc * tc);
    TA* dg = (TA*)a.dgates + (long)pb * K;
    dg[j] = from_f32<TA>(dc * g_ * i_ * (1.f - i_));
    dg[a.H + j] = from_f32<TA>(dc * p_cp * f_ * (1.f - f_));
    dg[2 * a.H + j] = from_f32<TA>(dc * i_ * (1.f - g_ * g_));
    dg[3 * a.H + j] = from_f32<TA>(dh * tc * o_ * (1.f - o_));
    a.dc_state[pbh] = dc * f_;
  }
}

static int g_step_dbg = 0;      // gic_debug_decoder_step: 1 | 2 | 4 load-ablation bits of tools/rollout_bench.py

// dynamic LDS beyond 64 KB: per kernel and per device (common.h grant_lds)
template <typename K>
int allow_lds(K kernel, size_t bytes, LdsGrant& g) {
  if (grant_lds(kernel, bytes, g)) return GIC_OK;
  set_last_error("decoder step kernel: cannot reserve %zu bytes of LDS", bytes);
  return GIC_ERR_LAUNCH;
}

// K chunk (elements) staged per pass: 1 KiB of a row in vocab_step (two tiles), 2 KiB in lstm_step (one tile); multiples of 32
int vocab_chunk(int dtype, int H) { const int cap = 1024 / dtype_size(dtype); const int h = (H + 31) & ~31; return h < cap ? h : cap; }
int lstm_chunk(int dtype, int ldx) { const int cap = 2048 / dtype_size(dtype); const int k = (ldx + 31) & ~31; return k < cap ? k : cap; }
size_t vocab_lds_bytes(int dtype, int H) {
  const size_t stage = (size_t)2 * kStepRows * ((size_t)vocab_chunk(dtype, H) * dtype_size(dtype) + 16);
  const size_t scratch = 8 * 2 * 64 * 16 + 4 * 4 * kStepRows * 4;
  return stage > scratch ? stage : scratch;
}
size_t lstm_lds_bytes(int dtype, int ldx) {
  const size_t stage = (size_t)kStepRows * ((size_t)lstm_chunk(dtype, ldx) * dtype_size(dtype) + 16);
  const size_t red = (size_t)8 * kStepRows * 17 * 4;
  return stage > red ? stage : red;
}

}  // namespace

bool decoder_step_supported(int dtype, int V, int E, int H, int NL) {
  static const bool off = getenv("GIC_NO_FUSED_ROLLOUT") != nullptr;
  if (off) return false;
  if (V < 4 || V % 4 || E % 8 || H % 8 || NL < 1) return false;
  (void)dtype;
  return (V + kVocabTile - 1) / kVocabTile <= 1024;                    // sample_finish's scale table
}

size_t decoder_step_part_floats(int B, int L, int V) {
  // + the 64-bit row keys (8-byte aligned) + two reserved 32-bit words behind them (ABI v2 sizing, unused)
  return (size_t)2 * L * B * ((V + kVocabTile - 1) / kVocabTile) + (size_t)2 * L * B + 4;
}

void decoder_step_debug(int v) { g_step_dbg = v; }

int decoder_step_max_rows() {
  static const int rows = [] { const char* e = getenv("GIC_FUSED_ROLLOUT_MAX_ROWS"); return e ? atoi(e) : 512; }();
  return rows;
}

int lstm_step(const LstmStepArgs& a, int dtype, hipStream_t stream) {
  GIC_CHECK_ARG(a.xh_t && a.xh_next && a.wcat && a.bsum && a.c_prev && a.c_new, "lstm_step: null buffer");
  GIC_CHECK_ARG(a.B > 0 && a.H > 0 && a.din > 0 && a.ldx == (long)a.din + a.H && a.ldx % 8 == 0 && a.din % 8 == 0 && a.gw % 8 == 0 && a.gw <= a.din,
                "lstm_step: bad dims");
  GIC_CHECK_ARG(!a.gather || (a.embed && a.V > 0 && (a.rowkey || a.force_ids)), "lstm_step: bad gather arguments");
  const dim3 grid((unsigned)cdiv(a.H, kUnitsPerBlock), (unsigned)cdiv(a.B, kStepRows));
  LstmStepArgs b = a;
  b.dbg = g_step_dbg;
  const int KC = lstm_chunk(dtype, (int)a.ldx);
  const size_t lds = lstm_lds_bytes(dtype, (int)a.ldx);
  if (dtype == DT_F32) {
    static LdsGrant g32;
    GIC_PROPAGATE(allow_lds(lstm_step_kernel<float>, lds, g32));
    hipLaunchKernelGGL((lstm_step_kernel<float>), grid, dim3(512), lds, stream, b, KC);
  } else {
    static LdsGrant g16;
    GIC_PROPAGATE(allow_lds(lstm_step_kernel<bf16_t>, lds, g16));
    hipLaunchKernelGGL((lstm_step_kernel<bf16_t>), grid, dim3(512), lds, stream, b, KC);
  }
  GIC_CHECK_LAUNCH("lstm_step");
  return GIC_OK;
}

int lstm_bwd_step(const LstmBwdStepArgs& a, int dtype, hipStream_t stream) {
  GIC_CHECK_ARG(a.gates && a.c_prev && a.c_cur && a.dc_state && a.dgates && a.B > 0 && a.H > 0 && a.H % 2 == 0, "lstm_bwd_step: bad arguments");
  GIC_CHECK_ARG((!a.dg_next || a.w_rec) && (!a.dg_up || a.w_up), "lstm_bwd_step: a gradient operand without its weights");
  const dim3 grid((unsigned)cdiv(a.H, 16), (unsigned)cdiv(a.B, 16));
  if (dtype == DT_F32) hipLaunchKernelGGL((lstm_bwd_step_kernel<float, 4>), grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((lstm_bwd_step_kernel<bf16_t, 8>), grid, dim3(512), 0, stream, a);
  GIC_CHECK_LAUNCH("lstm_bwd_step");
  return GIC_OK;
}

int vocab_step(const VocabStepArgs& a0, int dtype, hipStream_t stream) {
  VocabStepArgs a = a0;
  a.dbg = g_step_dbg;
  GIC_CHECK_ARG(a.h && a.wout && a.bias && a.part_m && a.part_s && a.rowkey, "vocab_step: null buffer");
  GIC_CHECK_ARG(a.B > 0 && a.V >= 4 && a.V % 4 == 0 && a.H % 8 == 0 && a.ldh % 8 == 0, "vocab_step: bad dims");
  GIC_CHECK_ARG(a.nblk == cdiv(a.V, kVocabTile), "vocab_step: nblk must be ceil(V / %d)", kVocabTile);
  GIC_CHECK_ARG(!a.out || a.out_stride % 4 == 0, "vocab_step: out row stride must be a multiple of 4");
  const size_t lds = vocab_lds_bytes(dtype, a.H);
  const int KC = vocab_chunk(dtype, a.H);
  const dim3 grid((unsigned)a.nblk, (unsigned)cdiv(a.B, kStepRows));
  if (dtype == DT_F32) {
    static LdsGrant g32;
    GIC_PROPAGATE(allow_lds(vocab_step_kernel<float, false>, lds, g32));
    hipLaunchKernelGGL((vocab_step_kernel<float, false>), grid, dim3(512), lds, stream, a, KC);
  } else {
    static LdsGrant g16;
    GIC_PROPAGATE(allow_lds(vocab_step_kernel<bf16_t, true>, lds, g16));
    hipLaunchKernelGGL((vocab_step_kernel<bf16_t, true>), grid, dim3(512), lds, stream, a, KC);
  }
  GIC_CHECK_LAUNCH("vocab_step");
  return GIC_OK;
}

int sample_finish(const SampleFinishArgs& a, int dtype, hipStream_t stream) {
  GIC_CHECK_ARG(a.part_m && a.part_s && a.rowkey && a.ids && a.nblk > 0 && a.nblk <= 1024, "sample_finish: bad partials");
  GIC_CHECK_ARG(a.B > 0 && a.L > 0 && a.V % 4 == 0, "sample_finish: bad dims");
  GIC_CHECK_ARG(!a.xh0 || a.embed, "sample_finish: the x rows need the embedding table");
  const dim3 grid((unsigned)((long)a.B * a.L));
  if (dtype == DT_F32) hipLaunchKernelGGL((sample_finish_kernel<float>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((sample_finish_kernel<bf16_t>), grid, dim3(256), 0, stream, a);
  GIC_CHECK_LAUNCH("sample_finish");
  return GIC_OK;
}

}  // namespace gic
#ifdef GIC_STAMPS
extern "C" int gic_debug_decoder_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gic::g_dstamp), sizeof(unsigned long long) * 2 * 2 * 8 * 2);
}
#endif
