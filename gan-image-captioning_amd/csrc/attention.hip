// Visual-attention caption decoder (BASELINE.json config 4; NO reference counterpart -- definition and CPU oracle in
// oracle/cpu_attention.py): the reference's roll-out loop (src/generator.py:55-81) with a Show-Attend-Tell soft attention over the
// trunk's feature map in front of the LSTM.
//
//   fp = fmap W_f^T + b_f                         one MFMA product per call              [B*P, A]
//   per step t:   attn_fwd     hp = W_h h_{t-1};  e_i = w_a . tanh(fp_i + hp);  alpha = softmax_i(e);  z = sum_i alpha_i fmap_i
//                 lstm_step    gates on [x_t | z_t | h_{t-1}]  (decoder_step.hip; x_t gathered from the embedding table)
//                 vocab_step   logits, Gumbel, softmax partials
//   backward, per step (reverse): lstm_bwd_step -> dz_t = dgates_t W_z (MFMA product) -> attn_bwd (softmax / tanh backward,
//                 d fp accumulated over steps, d hp, and dh_{t-1} += W_h^T d hp fed to the next lstm_bwd_step)
//   then batched over all steps: weight gradients (LSTM, W_h, W_f), input gradients of x (embedding scatter, d features).
//
// attn_fwd / attn_bwd: one workgroup per caption.  The P feature rows of a caption (P x C, 200 KB in bf16 at 7x7x2048) are
// streamed with coalesced 16-byte loads; the energies are reduced with wavefront shuffles (one position per wave at a time), the
// softmax over the P <= 1024 positions is a block reduction.
#include <stdlib.h>

#include "../../include/gicap.h"
#include "decoder_step.h"
#include "kernels.h"

namespace gic {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;

// 16 bytes of compute-dtype values as floats: NV = 8 (bf16) or 4 (f32)
template <typename TA> struct Vec16;
template <> struct Vec16<bf16_t> {
  static constexpr int NV = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 x = __builtin_bit_cast(bf16x8, *(gptr_u4)p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16_t)v[i];
    *(bf16x8*)p = x;
  }
};
template <> struct Vec16<float> {
  static constexpr int NV = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 x = *(const __attribute__((address_space(1))) f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = x[i];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
  }
};

constexpr int kAttnMaxP = 1024;      // positions per caption (LDS tables)
constexpr int kAttnMaxA = 2048;      // attention width (LDS table)

struct AttnFwdArgs {
  const void* fproj;                 // act [B, P, A]
  const float* w_a;                  // [A]
  const void* fmap;                  // act [B, P, C]
  float* alpha;                      // [B, P] out (saved)
  const float* hproj;                // [B, A] = h_{t-1} W_h^T (saved)
  void* z; long ld_z;                // act: z_t rows (XH_t + E)
  int P, A, H, C;
};

template <typename TA>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnFwdArgs a) {
  constexpr int NV = Vec16<TA>::NV;
  extern __shared__ float af_smem[];
  float* hp_s = af_smem;                   // [A]
  float* e_s = hp_s + a.A;                 // [P]
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // hp = W_h h_{t-1} was formed for all captions by one MFMA product (the launcher below): this block's row
  for (int j = tid; j < a.A; j += 256) hp_s[j] = a.hproj[(long)b * a.A + j];
  __syncthreads();
  // e_i = w_a . tanh(fp_i + hp)  -- one wave per position, kFwdPos positions' loads in flight at once (a load -> tanh -> reduce loop
  // pays one L2 round trip per position: 13 of them per wave at 49 positions)
  const TA* fp = (const TA*)a.fproj + (long)b * a.P * a.A;
  constexpr int kFwdPos = 4;
  for (int i0 = w; i0 < a.P; i0 += 4 * kFwdPos) {
    float s[kFwdPos];
#pragma unroll
    for (int u = 0; u < kFwdPos; ++u) s[u] = 0.f;
    for (int j0 = lane * NV; j0 < a.A; j0 += 64 * NV) {
      float v[kFwdPos][NV];
#pragma unroll
      for (int u = 0; u < kFwdPos; ++u) {                              // unconditional loads (positions past P re-read the wave's first)
        const int i = i0 + 4 * u < a.P ? i0 + 4 * u : i0;
        Vec16<TA>::load(fp + (long)i * a.A + j0, v[u]);
      }
      float wa[NV], hp[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) { wa[q] = a.w_a[j0 + q]; hp[q] = hp_s[j0 + q]; }
#pragma unroll
      for (int u = 0; u < kFwdPos; ++u)
#pragma unroll
        for (int q = 0; q < NV; ++q) s[u] += wa[q] * tanhf(v[u][q] + hp[q]);
    }
#pragma unroll
    for (int u = 0; u < kFwdPos; ++u) {
      const float t = wave_sum(s[u]);
      if (lane == 0 && i0 + 4 * u < a.P) e_s[i0 + 4 * u] = t;
    }
  }
  __syncthreads();
  // alpha = softmax over the P positions
  float m = -INFINITY;
  for (int i = tid; i < a.P; i += 256) m = fmaxf(m, e_s[i]);
  m = block_max(m, red);
  float s = 0.f;
  for (int i = tid; i < a.P; i += 256) s += expf(e_s[i] - m);
  s = block_sum(s, red);
  __syncthreads();
  for (int i = tid; i < a.P; i += 256) {
    const float al = expf(e_s[i] - m) / s;
    e_s[i] = al;
    a.alpha[(long)b * a.P + i] = al;
  }
  __syncthreads();
  // z = sum_i alpha_i fmap_i  -- threads over 16-byte pieces of the feature row (coalesced), positions streamed
  const TA* fm = (const TA*)a.fmap + (long)b * a.P * a.C;
  TA* zrow = (TA*)a.z + (long)b * a.ld_z;
  for (int c0 = tid * NV; c0 < a.C; c0 += 256 * NV) {
    float acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q] = 0.f;
    // 8 positions' feature pieces in flight per thread (the positions of a caption are a dependent chain only through `acc`)
    constexpr int kZ = 8;
    for (int i0 = 0; i0 < a.P; i0 += kZ) {
      float v[kZ][NV];
#pragma unroll
      for (int u = 0; u < kZ; ++u) Vec16<TA>::load(fm + (long)(i0 + u < a.P ? i0 + u : i0) * a.C + c0, v[u]);
#pragma unroll
      for (int u = 0; u < kZ; ++u) {
        const float al = i0 + u < a.P ? e_s[i0 + u] : 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[q] += al * v[u][q];
      }
    }
    Vec16<TA>::store(zrow + c0, acc);
  }
}

struct AttnBwdArgs {
  const float* dz;                   // [B, C]
  const float* alpha;                // [B, P]
  const float* hproj;                // [B, A]
  const void* fproj;                 // act [B, P, A]
  const void* fmap;                  // act [B, P, C]
  const float* w_a;                  // [A]
  float* dalpha;                     // [B, P] scratch: dz . fmap_i
  float* dfproj;                     // [B, P, A] accumulated over the steps (zeroed by the caller)
  void* dhproj;                      // act [B, A] of this step (operand of the W_h weight gradient and of dh_{t-1} += dhp W_h)
  float* dwa_rows;                   // [B, A] accumulated over the steps (zeroed by the caller): d w_a per caption
  float* dz_zero;                    // [B, C] or null: zeroed by attn_bwd after its last reader (the next step's split-K product adds into it)
  int P, A, H, C;
};

// d alpha[b, i] = dz[b, :] . fmap[b, i, :]  -- one wave per (caption, position): B*P/4 blocks, coalesced 16-byte feature loads
template <typename TA>
__global__ __launch_bounds__(256) void attn_dalpha_kernel(const AttnBwdArgs a, int B) {
  constexpr int NV = Vec16<TA>::NV;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // b * P + i
  if (row >= (long)B * a.P) return;
  const int b = (int)(row / a.P);
  const TA* f = (const TA*)a.fmap + row * a.C;
  const float* dz = a.dz + (long)b * a.C;
  float s = 0.f;
  for (int c0 = lane * NV; c0 < a.C; c0 += 64 * NV) {
    float v[NV];
    Vec16<TA>::load(f + c0, v);
#pragma unroll
    for (int q = 0; q < NV; ++q) s += v[q] * dz[c0 + q];
  }
  s = wave_sum(s);
  if (lane == 0) a.dalpha[row] = s;
}

// softmax backward + tanh backward for a slice of 64*NV attention columns of one caption (grid = A/(64 NV) x B): de is recomputed per
// block from d alpha (P values); each of the 4 waves sweeps every 4th position over the block's columns, one 16-byte piece per lane;
// d hp of the slice (sum over ALL positions) is complete inside the block, d fp rows are this caption's own (no atomics).
template <typename TA>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdArgs a) {
  constexpr int NV = Vec16<TA>::NV;
  extern __shared__ float ab_smem[];
  float* de_s = ab_smem;                   // [P]
  float* dhp_s = de_s + a.P;               // [4][64*NV]
  float* dwa_s = dhp_s + 4 * 64 * NV;      // [4][64*NV]
  __shared__ float red[16];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j0 = blockIdx.x * 64 * NV + lane * NV;                     // this lane's columns
  float dot = 0.f;
  for (int i = tid; i < a.P; i += 256) dot += a.alpha[(long)b * a.P + i] * a.dalpha[(long)b * a.P + i];
  dot = block_sum(dot, red);
  for (int i = tid; i < a.P; i += 256) de_s[i] = a.alpha[(long)b * a.P + i] * (a.dalpha[(long)b * a.P + i] - dot);
  __syncthreads();
  float dhp[NV], dwa[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) { dhp[q] = 0.f; dwa[q] = 0.f; }
  if (j0 < a.A) {
    const TA* fp = (const TA*)a.fproj + (long)b * a.P * a.A;
    float* dfp = a.dfproj + (long)b * a.P * a.A;
    float hp[NV], wa[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) { hp[q] = a.hproj[(long)b * a.A + j0 + q]; wa[q] = a.w_a[j0 + q]; }
    // kBwdPos positions per pass: their fp pieces and d fp rows (read-modify-write) are all requested before the first use
    constexpr int kBwdPos = 4;
    for (int i0 = w; i0 < a.P; i0 += 4 * kBwdPos) {
      float v[kBwdPos][NV], d[kBwdPos][NV];
#pragma unroll
      for (int u = 0; u < kBwdPos; ++u) {
        const int i = i0 + 4 * u < a.P ? i0 + 4 * u : i0;
        Vec16<TA>::load(fp + (long)i * a.A + j0, v[u]);
#pragma unroll
        for (int q = 0; q < NV; q += 4) *(f32x4*)&d[u][q] = *(const f32x4*)(dfp + (long)i * a.A + j0 + q);
      }
#pragma unroll
      for (int u = 0; u < kBwdPos; ++u) {
        const int i = i0 + 4 * u;
        if (i >= a.P) break;
        const float de = de_s[i];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const float th = tanhf(v[u][q] + hp[q]);
          const float dpre = de * wa[q] * (1.f - th * th);
          d[u][q] += dpre;
          dhp[q] += dpre;
          dwa[q] += de * th;
        }
#pragma unroll
        for (int q = 0; q < NV; q += 4) *(f32x4*)(dfp + (long)i * a.A + j0 + q) = *(const f32x4*)&d[u][q];
      }
    }
  }
  if (a.dz_zero && blockIdx.x == 0)                        // dz of this step has been consumed by attn_dalpha (an earlier launch)
    for (int c = tid * 4; c < a.C; c += 256 * 4) *(f32x4*)(a.dz_zero + (long)b * a.C + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NV; ++q) { dhp_s[(w * 64 + lane) * NV + q] = dhp[q]; dwa_s[(w * 64 + lane) * NV + q] = dwa[q]; }
  __syncthreads();
  for (int c = tid; c < 64 * NV; c += 256) {
    const int j = blockIdx.x * 64 * NV + c;
    if (j < a.A) {
      const float sh = dhp_s[c] + dhp_s[64 * NV + c] + dhp_s[2 * 64 * NV + c] + dhp_s[3 * 64 * NV + c];
      const float sw = dwa_s[c] + dwa_s[64 * NV + c] + dwa_s[2 * 64 * NV + c] + dwa_s[3 * 64 * NV + c];
      ((TA*)a.dhproj)[(long)b * a.A + j] = from_f32<TA>(sh);
      a.dwa_rows[(long)b * a.A + j] += sw;                             // this block owns (b, j): plain accumulate over the steps
    }
  }
}

struct ACtx {
  int B, L, V, E, H, C, P, A, dt;
  long ldx() const { return (long)E + C + H; }
  int din() const { return E + C; }
  size_t asz() const { return (size_t)dtype_size(dt); }
};

int check_attn_dims(const gic_attn_dims* d, ACtx& c) {
  GIC_CHECK_ARG(d, "attn: null dims");
  GIC_CHECK_ARG(d->B > 0 && d->L > 0 && d->V >= 4 && d->E > 0 && d->H > 0 && d->C > 0 && d->P > 0 && d->A > 0, "attn: bad dims");
  GIC_CHECK_ARG(d->dtype == DT_F32 || d->dtype == DT_BF16, "attn: bad dtype");
  GIC_CHECK_ARG(d->V % 4 == 0 && d->E % 8 == 0 && d->H % 8 == 0 && d->C % 8 == 0 && d->A % 8 == 0,
                "attn: V must be a multiple of 4, E / H / C / A multiples of 8 (16-byte pieces; pad the vocabulary / widths)");
  GIC_CHECK_ARG(d->P <= kAttnMaxP && d->A <= kAttnMaxA, "attn: at most %d positions and attention width %d", kAttnMaxP, kAttnMaxA);
  c = ACtx{d->B, d->L, d->V, d->E, d->H, d->C, d->P, d->A, d->dtype};
  return GIC_OK;
}

template <typename TA>
int attn_fwd_t(const ACtx& c, const gic_attn_params* P, const gic_attn_shadow* S, const gic_attn_state* st, const float* features,
               const void* fmap, const float* noise_u, uint64_t seed, float temperature, int pretrain, void* out, int64_t* ids,
               const float* h0, const float* c0, const float* t_dev, const uint64_t* seed_dev, hipStream_t stream) {
  const int B = c.B, L = c.L, V = c.V, E = c.E, H = c.H;
  const long ld = c.ldx();
  // slot 0: initial (h, c) -- zeros, or the caller's states (sample(features, fmap, states=(h0, c0))) -- and features -> x_0
  GIC_PROPAGATE(fill_zero(st->xh, (size_t)B * ld * c.asz(), stream));
  if (h0) GIC_PROPAGATE(cast2d(h0, DT_F32, H, (char*)st->xh + (size_t)(ld - H) * c.asz(), c.dt, ld, B, H, stream));
  if (c0) GIC_PROPAGATE(cast2d(c0, DT_F32, H, st->c, DT_F32, H, B, H, stream));
  else GIC_PROPAGATE(fill_zero(st->c, (size_t)B * H * sizeof(float), stream));
  GIC_PROPAGATE(cast2d(features, DT_F32, E, st->xh, c.dt, ld, B, E, stream));
  {  // fp = fmap W_f^T + b_f
    GemmDesc g;
    g.A = fmap; g.lda = c.C; g.B = S->wf; g.ldb = c.C; g.C = st->fproj; g.ldc = c.A;
    g.M = B * c.P; g.N = c.A; g.K = c.C; g.in_dtype = c.dt; g.out_dtype = c.dt; g.bias = P->b_f;
    GIC_PROPAGATE(gemm(g, stream));
  }
  const int nblk = cdiv(V, kVocabTile);
  const long per = (long)L * B * nblk;
  float* part_m = st->part;
  float* part_s = st->part + per;
  unsigned long long* rowkey = (unsigned long long*)(st->part + ((2 * per + 1) & ~1l));
  GIC_PROPAGATE(fill_zero(rowkey, (size_t)L * B * sizeof(unsigned long long), stream));
  const size_t lds = (size_t)(c.A + c.P) * sizeof(float);
  static LdsGrant granted;
  if (!grant_lds(attn_fwd_kernel<TA>, lds, granted)) {
    set_last_error("attention kernel: cannot reserve %zu bytes of LDS", lds);
    return GIC_ERR_LAUNCH;
  }
  // the per-step products below are split-K (f32 atomics into a zeroed C): ONE fill for all L steps instead of a zeroing launch per step
  GIC_PROPAGATE(fill_zero(st->hproj, (size_t)L * B * c.A * sizeof(float), stream));
  for (int t = 0; t < L; ++t) {
    TA* xh_t = (TA*)st->xh + (long)t * B * ld;
    {  // hp [B, A] = h_{t-1} W_h^T for all captions: one product
      GemmDesc g;
      g.A = xh_t + c.din(); g.lda = ld; g.B = S->wh; g.ldb = H; g.C = st->hproj + (long)t * B * c.A; g.ldc = c.A;
      g.M = B; g.N = c.A; g.K = H; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.c_zeroed = 1;
      GIC_PROPAGATE(gemm(g, stream));
    }
    AttnFwdArgs f;
    f.fproj = st->fproj; f.w_a = P->w_a; f.fmap = fmap;
    f.alpha = st->alpha + (long)t * B * c.P; f.hproj = st->hproj + (long)t * B * c.A; f.z = xh_t + E; f.ld_z = ld;
    f.P = c.P; f.A = c.A; f.H = H; f.C = c.C;
    hipLaunchKernelGGL((attn_fwd_kernel<TA>), dim3(B), dim3(256), lds, stream, f);
    GIC_CHECK_LAUNCH("attn_fwd");
    LstmStepArgs a;
    a.xh_t = xh_t; a.xh_next = xh_t + (long)B * ld; a.wcat = S->wcat; a.bsum = S->bsum;
    a.c_prev = st->c + (long)t * B * H; a.c_new = st->c + (long)(t + 1) * B * H;
    a.gates = st->gates + (long)t * B * 4 * H;
    a.h_out = (TA*)st->hout + (long)t * H; a.ld_out = (long)L * H;
    a.B = B; a.H = H; a.din = c.din(); a.ldx = ld; a.gw = E;
    if (t > 0) { a.gather = 1; a.embed = P->embed; a.V = V; a.rowkey = rowkey + (long)(t - 1) * B; a.tprev = t - 1; }
    GIC_PROPAGATE(lstm_step(a, c.dt, stream));
    VocabStepArgs v;
    v.h = xh_t + (long)B * ld + c.din(); v.ldh = ld; v.wout = S->wout; v.bias = P->b_out;
    v.u = noise_u ? noise_u + (long)t * B * V : nullptr;
    v.seed = seed; v.rng_stream = (uint64_t)t; v.temperature = temperature; v.pretrain = pretrain;
    v.t_dev = t_dev; v.seed_dev = seed_dev;
    v.out = (TA*)out + (long)t * V; v.out_stride = (long)L * V;
    v.part_m = part_m + (long)t * B * nblk; v.part_s = part_s + (long)t * B * nblk; v.rowkey = rowkey + (long)t * B;
    v.nblk = nblk; v.B = B; v.V = V; v.H = H;
    GIC_PROPAGATE(vocab_step(v, c.dt, stream));
  }
  SampleFinishArgs f;
  f.part_m = part_m; f.part_s = part_s; f.rowkey = rowkey; f.nblk = nblk; f.B = B; f.L = L; f.V = V; f.E = E;
  f.pretrain = pretrain; f.out = out; f.ids = ids; f.embed = P->embed; f.xh0 = st->xh; f.ldx0 = ld;
  return sample_finish(f, c.dt, stream);
}

template <typename TA>
int attn_bwd_t(const ACtx& c, const gic_attn_params* P, const gic_attn_shadow* S, const gic_attn_state* st, const gic_attn_bwd_ws* ws,
               const void* fmap, const void* probs, const int64_t* ids, const void* d_out, float temperature, int pretrain,
               const gic_attn_grads* G, const float* t_dev, hipStream_t stream) {
  const int B = c.B, L = c.L, V = c.V, E = c.E, H = c.H, C = c.C, A = c.A;
  const long ld = c.ldx(), BL = (long)B * L;
  GIC_PROPAGATE(decoder_output_bwd(c.dt, B, L, V, H, probs, d_out, temperature, t_dev, pretrain, ws->dlogits, S->wout, st->hout, ws->dhout,
                                   G->w_out, G->b_out, stream));
  GIC_PROPAGATE(fill_zero(ws->dc, (size_t)B * H * sizeof(float), stream));
  GIC_PROPAGATE(fill_zero(ws->dfproj, (size_t)B * c.P * A * sizeof(float), stream));
  GIC_PROPAGATE(fill_zero(ws->dwa_rows, (size_t)B * A * sizeof(float), stream));
  // dz / dh_extra receive split-K products (f32 atomics) every step: zeroed here once, then re-zeroed by their last reader of the step
  // (attn_bwd / the next lstm_bwd_step) instead of by a launch of their own
  GIC_PROPAGATE(fill_zero(ws->dz, (size_t)B * C * sizeof(float), stream));
  GIC_PROPAGATE(fill_zero(ws->dh_extra, (size_t)B * H * sizeof(float), stream));
  constexpr int kCols = 64 * Vec16<TA>::NV;                       // attention columns per attn_bwd block
  const size_t lds = (size_t)(c.P + 8 * kCols) * sizeof(float);
  const TA* wt = (const TA*)S->wcat_t;                      // [ldx, 4H]: rows 0..E-1 x, E..E+C-1 z, E+C.. h
  for (int t = L - 1; t >= 0; --t) {
    LstmBwdStepArgs a;
    a.dh_above = ws->dhout + (long)t * H; a.ld_above = (long)L * H;
    if (t + 1 < L) { a.dg_next = (TA*)ws->dgates + (long)(t + 1) * B * 4 * H; a.w_rec = wt + (long)c.din() * 4 * H; a.dh_extra = ws->dh_extra; a.zero_extra = 1; }
    a.gates = st->gates + (long)t * B * 4 * H;
    a.c_prev = st->c + (long)t * B * H; a.c_cur = st->c + (long)(t + 1) * B * H;
    a.dc_state = ws->dc; a.dgates = (TA*)ws->dgates + (long)t * B * 4 * H; a.B = B; a.H = H;
    GIC_PROPAGATE(lstm_bwd_step(a, c.dt, stream));
    {  // dz_t [B, C] = dgates_t . W_z   (rows E .. E+C-1 of Wcat^T)
      GemmDesc g;
      g.A = a.dgates; g.lda = 4 * H; g.a_kc = 1; g.B = wt + (long)E * 4 * H; g.ldb = 4 * H; g.b_kc = 1;
      g.C = ws->dz; g.ldc = C; g.M = B; g.N = C; g.K = 4 * H; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.c_zeroed = 1;
      GIC_PROPAGATE(gemm(g, stream));
    }
    AttnBwdArgs f;
    f.dz = ws->dz; f.alpha = st->alpha + (long)t * B * c.P; f.hproj = st->hproj + (long)t * B * A; f.fproj = st->fproj; f.fmap = fmap;
    f.w_a = P->w_a; f.dalpha = ws->dalpha; f.dfproj = ws->dfproj; f.dhproj = (TA*)ws->dhproj + (long)t * B * A; f.dwa_rows = ws->dwa_rows;
    f.P = c.P; f.A = A; f.H = H; f.C = C;
    f.dz_zero = (C % 4 == 0) ? ws->dz : nullptr;
    hipLaunchKernelGGL((attn_dalpha_kernel<TA>), dim3((unsigned)cdiv((long)B * c.P, 4)), dim3(256), 0, stream, f, B);
    GIC_CHECK_LAUNCH("attn_dalpha");
    hipLaunchKernelGGL((attn_bwd_kernel<TA>), dim3((unsigned)cdiv(A, kCols), (unsigned)B), dim3(256), lds, stream, f);
    GIC_CHECK_LAUNCH("attn_bwd");
    if (t > 0) {  // dh_{t-1} += dhp_t W_h  (consumed by the next lstm_bwd_step as dh_extra)
      GemmDesc g;
      g.A = f.dhproj; g.lda = A; g.a_kc = 1; g.B = S->wh; g.ldb = H; g.b_kc = 0; g.C = ws->dh_extra; g.ldc = H;
      g.M = B; g.N = H; g.K = A; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.c_zeroed = 1;
      GIC_PROPAGATE(gemm(g, stream));
    }
  }
  // ---- batched over all steps
  {  // d x_t [L*B, E] = dgates W_x  (rows 0..E-1 of Wcat^T)
    GemmDesc g;
    g.A = ws->dgates; g.lda = 4 * H; g.a_kc = 1; g.B = wt; g.ldb = 4 * H; g.b_kc = 1; g.C = ws->dx; g.ldc = E;
    g.M = (int)BL; g.N = E; g.K = 4 * H; g.in_dtype = c.dt; g.out_dtype = DT_F32;
    GIC_PROPAGATE(gemm(g, stream));
  }
  GIC_PROPAGATE(cast2d(ws->dx, DT_F32, E, G->features, DT_F32, E, B, E, stream));
  GIC_PROPAGATE(embed_scatter_time(ws->dx, E, ids, G->embed, B, L, E, V, stream));
  {  // LSTM weight gradients over all L*B rows
    GemmDesc w;
    w.A = ws->dgates; w.lda = 4 * H; w.a_kc = 0; w.b_kc = 0; w.ldb = ld; w.M = 4 * H; w.K = (int)BL; w.in_dtype = c.dt; w.out_dtype = DT_F32;
    w.B = st->xh; w.N = c.din(); w.C = G->w_ih; w.ldc = c.din();
    GIC_PROPAGATE(gemm(w, stream));
    w.B = (const TA*)st->xh + c.din(); w.N = H; w.C = G->w_hh; w.ldc = H;
    GIC_PROPAGATE(gemm(w, stream));
    GIC_PROPAGATE(colsum(ws->dgates, c.dt, 4 * H, BL, 4 * H, G->b_ih, G->b_hh, 0, stream));
  }
  {  // dW_h [A, H] = sum_t dhp_t^T h_{t-1}
    GemmDesc w;
    w.A = ws->dhproj; w.lda = A; w.a_kc = 0; w.B = (const TA*)st->xh + c.din(); w.ldb = ld; w.b_kc = 0; w.C = G->w_h; w.ldc = H;
    w.M = A; w.N = H; w.K = (int)BL; w.in_dtype = c.dt; w.out_dtype = DT_F32;
    GIC_PROPAGATE(gemm(w, stream));
  }
  {  // dW_f [A, C] = d fp^T fmap ; db_f = colsum(d fp)
    const long rows = (long)B * c.P;
    const void* dfp = ws->dfproj;
    if (c.dt != DT_F32) {
      GIC_PROPAGATE(cast2d(ws->dfproj, DT_F32, A, ws->dfproj_act, c.dt, A, rows, A, stream));
      dfp = ws->dfproj_act;
    }
    GemmDesc w;
    w.A = dfp; w.lda = A; w.a_kc = 0; w.B = fmap; w.ldb = C; w.b_kc = 0; w.C = G->w_f; w.ldc = C;
    w.M = A; w.N = C; w.K = (int)rows; w.in_dtype = c.dt; w.out_dtype = DT_F32;
    GIC_PROPAGATE(gemm(w, stream));
    GIC_PROPAGATE(colsum(ws->dfproj, DT_F32, A, rows, A, G->b_f, nullptr, 0, stream));
  }
  return colsum(ws->dwa_rows, DT_F32, A, B, A, G->w_a, nullptr, 0, stream);
}

}  // namespace
}  // namespace gic

using namespace gic;

extern "C" {

int gic_attn_prepare(const gic_attn_dims* dims, const gic_attn_params* P, const gic_attn_shadow* S, void* stream_) {
  ACtx c;
  GIC_PROPAGATE(check_attn_dims(dims, c));
  GIC_CHECK_ARG(P && S && P->w_ih && P->w_hh && P->b_ih && P->b_hh && P->w_out && P->w_f && P->w_h, "attn_prepare: null parameter");
  GIC_CHECK_ARG(S->wcat && S->bsum && S->wout && S->wcat_t && S->wf && S->wh, "attn_prepare: null shadow buffer");
  hipStream_t stream = (hipStream_t)stream_;
  // Wcat = [w_ih | w_hh] through the LSTM decoder's prepare (E' = E + C inputs)
  gic_decoder_dims d = {1, 1, c.V, c.din(), c.H, 1, c.dt};
  gic_decoder_params dp = {};
  dp.embed = P->embed; dp.w_ih[0] = P->w_ih; dp.w_hh[0] = P->w_hh; dp.b_ih[0] = P->b_ih; dp.b_hh[0] = P->b_hh; dp.w_out = P->w_out; dp.b_out = P->b_out;
  gic_decoder_shadow ds = {};
  ds.wcat[0] = S->wcat; ds.bsum[0] = S->bsum; ds.wout = S->wout; ds.wcat_t[0] = S->wcat_t;
  GIC_PROPAGATE(gic_decoder_prepare(&d, &dp, &ds, stream_));
  GIC_PROPAGATE(cast2d(P->w_f, DT_F32, c.C, S->wf, c.dt, c.C, c.A, c.C, stream));
  return cast2d(P->w_h, DT_F32, c.H, S->wh, c.dt, c.H, c.A, c.H, stream);
}

int gic_attn_sample_fwd(const gic_attn_dims* dims, const gic_attn_params* P, const gic_attn_shadow* S, const gic_attn_state* st,
                        const float* features, const void* fmap, const float* noise_u, uint64_t seed, float temperature, int pretrain,
                        void* out, int64_t* ids, const float* h0, const float* c0, const gic_step_scalars* dev_scalars, int seed_slot,
                        void* stream) {
  ACtx c;
  GIC_PROPAGATE(check_attn_dims(dims, c));
  GIC_CHECK_ARG(!dev_scalars || (seed_slot >= 0 && seed_slot < GIC_STEP_SEEDS), "attn_sample_fwd: seed_slot out of range");
  const float* t_dev = dev_scalars ? &dev_scalars->temperature : nullptr;
  const uint64_t* seed_dev = dev_scalars ? &dev_scalars->seed[seed_slot] : nullptr;
  GIC_CHECK_ARG(P && S && st && features && fmap && out && ids, "attn_sample_fwd: null argument");
  GIC_CHECK_ARG(P->embed && P->b_out && P->b_f && P->w_a && S->wcat && S->bsum && S->wout && S->wf && S->wh, "attn_sample_fwd: null weights");
  GIC_CHECK_ARG(st->xh && st->gates && st->c && st->hout && st->part && st->fproj && st->alpha && st->hproj, "attn_sample_fwd: null state buffer");
  if (c.dt == DT_F32)
    return attn_fwd_t<float>(c, P, S, st, features, fmap, noise_u, seed, temperature, pretrain, out, ids, h0, c0, t_dev, seed_dev, (hipStream_t)stream);
  return attn_fwd_t<bf16_t>(c, P, S, st, features, fmap, noise_u, seed, temperature, pretrain, out, ids, h0, c0, t_dev, seed_dev, (hipStream_t)stream);
}

int gic_attn_sample_bwd(const gic_attn_dims* dims, const gic_attn_params* P, const gic_attn_shadow* S, const gic_attn_state* st,
                        const gic_attn_bwd_ws* ws, const void* fmap, const void* probs, const int64_t* ids, const void* d_out,
                        float temperature, int pretrain, const gic_attn_grads* G, const gic_step_scalars* dev_scalars, void* stream) {
  ACtx c;
  GIC_PROPAGATE(check_attn_dims(dims, c));
  const float* t_dev = dev_scalars ? &dev_scalars->temperature : nullptr;
  GIC_CHECK_ARG(P && S && st && ws && fmap && probs && ids && d_out && G, "attn_sample_bwd: null argument");
  GIC_CHECK_ARG(S->wcat_t && S->wout && S->wh && P->w_a, "attn_sample_bwd: null weights");
  GIC_CHECK_ARG(ws->dlogits && ws->dhout && ws->dgates && ws->dc && ws->dz && ws->dalpha && ws->dh_extra && ws->dhproj && ws->dfproj && ws->dwa_rows && ws->dx &&
                (c.dt == DT_F32 || ws->dfproj_act), "attn_sample_bwd: null workspace buffer");
  GIC_CHECK_ARG(G->embed && G->w_ih && G->w_hh && G->b_ih && G->b_hh && G->w_out && G->b_out && G->w_f && G->b_f && G->w_h && G->w_a && G->features,
                "attn_sample_bwd: null gradient buffer");
  if (c.dt == DT_F32)
    return attn_bwd_t<float>(c, P, S, st, ws, fmap, probs, ids, d_out, temperature, pretrain, G, t_dev, (hipStream_t)stream);
  return attn_bwd_t<bf16_t>(c, P, S, st, ws, fmap, probs, ids, d_out, temperature, pretrain, G, t_dev, (hipStream_t)stream);
}

}  // extern "C"
