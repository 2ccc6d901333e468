// Generator hot loop: Decoder.sample forward / backward (reference src/generator.py:55-96).
//
// Layout (DESIGN.md "decoder"): the recurrent side is time-major -- per layer one
// XH buffer [(L+1), B, Din+H] holding [LSTM input | previous hidden] so that step t's
// gate GEMM reads one contiguous [B, Din+H] operand and the batched wgrad reads one
// [L*B, Din+H] operand; the vocabulary side (probs, d_probs, d_logits, hout) is
// batch-major [B, L, .] like the tensors the reference returns.  The pointwise kernels
// bridge the two with explicit strides.
#include <stdlib.h>

#include "../../include/gicap.h"
#include "decoder_step.h"
#include "kernels.h"

namespace gic {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---- LSTM pointwise forward: gates pre-activation [B,4H] (i,f,g,o blocks) -> gates, c, h (3 destinations)
template <typename TA>
__global__ void lstm_pointwise_fwd_kernel(const float* __restrict__ gpre, const float* __restrict__ c_prev,
                                          float* __restrict__ gates, float* __restrict__ c_new,
                                          TA* __restrict__ h_next, long ld_next,      // XH_l[t+1][:, Din:]
                                          TA* __restrict__ h_up, long ld_up,          // XH_{l+1}[t][:, :H] or null
                                          TA* __restrict__ h_out, long ld_out,        // hout[b, t, :] or null
                                          int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  const float* g = gpre + (long)b * 4 * H;
  const float i_ = sigmoidf_(g[j]);
  const float f_ = sigmoidf_(g[H + j]);
  const float g_ = tanhf(g[2 * H + j]);
  const float o_ = sigmoidf_(g[3 * H + j]);
  const float c = f_ * c_prev[idx] + i_ * g_;
  const float h = o_ * tanhf(c);
  if (gates) {
    float* go = gates + (long)b * 4 * H;
    go[j] = i_; go[H + j] = f_; go[2 * H + j] = g_; go[3 * H + j] = o_;
  }
  c_new[idx] = c;
  h_next[(long)b * ld_next + j] = from_f32<TA>(h);
  if (h_up) h_up[(long)b * ld_up + j] = from_f32<TA>(h);
  if (h_out) h_out[(long)b * ld_out + j] = from_f32<TA>(h);
}

// ---- LSTM pointwise backward for one (layer, step)
template <typename TA>
__global__ void lstm_pointwise_bwd_kernel(const float* __restrict__ dh_a, long ld_a,   // d h from above (dhout[b,t,:] or upper layer dx)
                                          const float* __restrict__ dh_b, long ld_b,   // recurrent d h from step t+1 (or null)
                                          const float* __restrict__ gates, const float* __restrict__ c_prev,
                                          const float* __restrict__ c_cur, float* __restrict__ dc_state,
                                          TA* __restrict__ dgates, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  float dh = dh_a[(long)b * ld_a + j];
  if (dh_b) dh += dh_b[(long)b * ld_b + j];
  const float* g = gates + (long)b * 4 * H;
  const float i_ = g[j], f_ = g[H + j], g_ = g[2 * H + j], o_ = g[3 * H + j];
  const float tc = tanhf(c_cur[idx]);
  const float dc = dc_state[idx] + dh * o_ * (1.f - tc * tc);
  TA* dg = dgates + (long)b * 4 * H;
  dg[j] = from_f32<TA>(dc * g_ * i_ * (1.f - i_));
  dg[H + j] = from_f32<TA>(dc * c_prev[idx] * f_ * (1.f - f_));
  dg[2 * H + j] = from_f32<TA>(dc * i_ * (1.f - g_ * g_));
  dg[3 * H + j] = from_f32<TA>(dh * tc * o_ * (1.f - o_));
  dc_state[idx] = dc * f_;
}

// ---- per-row Gumbel-softmax + argmax + next-input embedding gather (generator.py:68-76, 84-96)
// One 256-thread block per batch row.  y = (o + g) * T is written back over the logits scratch,
// then exp / sum / normalise; argmax over the stored probabilities with first-index tie-break.
template <typename TA>
__global__ __launch_bounds__(256) void gumbel_softmax_argmax_kernel(
    float* __restrict__ logits, const float* __restrict__ u, uint64_t seed, uint64_t rng_stream, float temperature,
    int pretrain, TA* __restrict__ out, long out_row_stride, int64_t* __restrict__ ids, long ids_stride,
    const float* __restrict__ embed, TA* __restrict__ x_next, long ld_x, int V, int E,
    const int64_t* __restrict__ force_ids, const int32_t* __restrict__ force_len, int t) {
  __shared__ float red[16];
  __shared__ int red_i[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* row = logits + (long)b * V;
  TA* orow = out ? out + (long)b * out_row_stride : nullptr;
  const float eps = 1e-10f;

  // pass 1: y, row max
  float mx = -INFINITY;
  for (int v = tid; v < V; v += 256) {
    float y = row[v];
    if (!pretrain) {
      float uu;
      if (u) {
        uu = u[(long)b * V + v];
      } else {
        uint32_t r[4];
        const uint64_t idx = (uint64_t)b * (uint64_t)V + (uint64_t)v;
        Philox::gen(seed, rng_stream, idx >> 2, r);
        uu = Philox::u01(r[idx & 3]);
      }
      const float g = -logf(-logf(uu + eps) + eps);
      y = (y + g) * temperature;
      row[v] = y;
    }
    mx = fmaxf(mx, y);
  }
  mx = block_max(mx, red);
  // pass 2: exp, sum
  float sm = 0.f;
  for (int v = tid; v < V; v += 256) sm += expf(row[v] - mx);
  sm = block_sum(sm, red);
  // pass 3: probabilities, argmax (first maximal index), output
  float best = -1.f;
  int best_i = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float y = row[v];
    const float p = expf(y - mx) / sm;
    if (orow) orow[v] = from_f32<TA>(pretrain ? y : p);
    if (p > best) { best = p; best_i = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  __syncthreads();                      // block_sum's readers are done with red[]
  if ((tid & 63) == 0) { red[tid >> 6] = best; red_i[tid >> 6] = best_i; }
  __syncthreads();
  best = red[0]; best_i = red_i[0];
  for (int w = 1; w < 4; ++w)
    if (red[w] > best || (red[w] == best && red_i[w] < best_i)) { best = red[w]; best_i = red_i[w]; }
  if (force_ids && (!force_len || t < force_len[b])) {       // forced trajectory (gicap.h, gic_decoder_sample_opts)
    const long f = force_ids[(long)b * ids_stride];
    best_i = (int)(f < 0 ? 0 : (f >= V ? V - 1 : f));
  }
  if (tid == 0) ids[(long)b * ids_stride] = best_i;
  if (x_next)
    for (int e = tid; e < E; e += 256) x_next[(long)b * ld_x + e] = from_f32<TA>(embed[(long)best_i * E + e]);
}

// ---- register-resident variant for V % 4 == 0, V <= 4096*QPT: one 1024-thread block per row, each thread owns
// QPT quads of 4 consecutive vocabulary entries (16-B loads of logits / u, one Philox4x32 call per quad), the row
// is read once and never re-read; reductions = wavefront shuffles + one LDS hop.  FAST selects the hardware
// exp/log approximations (bf16 compute mode); the f32 parity mode keeps libm-accurate logf/expf.
template <typename TA, int QPT, bool FAST>
__global__ __launch_bounds__(1024) void gumbel_softmax_argmax_reg_kernel(
    const float* __restrict__ logits, const float* __restrict__ u, uint64_t seed, uint64_t rng_stream, float temperature,
    int pretrain, TA* __restrict__ out, long out_row_stride, int64_t* __restrict__ ids, long ids_stride,
    const float* __restrict__ embed, TA* __restrict__ x_next, long ld_x, int V, int E,
    const int64_t* __restrict__ force_ids, const int32_t* __restrict__ force_len, int t) {
  __shared__ float red[16];
  __shared__ int red_i[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (long)b * V;
  TA* orow = out ? out + (long)b * out_row_stride : nullptr;
  const float eps = 1e-10f;
  const int nq = V >> 2;
  float y[QPT][4];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < QPT; ++e) {
    const int q = tid + e * 1024;
    if (q < nq) {
      const float4 lv = *(const float4*)(row + 4 * q);
      float yy[4] = {lv.x, lv.y, lv.z, lv.w};
      if (!pretrain) {
        float uu[4];
        if (u) {
          const float4 uv = *(const float4*)(u + (long)b * V + 4 * q);
          uu[0] = uv.x; uu[1] = uv.y; uu[2] = uv.z; uu[3] = uv.w;
        } else {
          uint32_t r0, r1, r2, r3;
          Philox::gen4(seed, rng_stream, (uint64_t)b * (uint64_t)nq + (uint64_t)q, r0, r1, r2, r3);
          uu[0] = Philox::u01(r0); uu[1] = Philox::u01(r1); uu[2] = Philox::u01(r2); uu[3] = Philox::u01(r3);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float g = FAST ? -__logf(-__logf(uu[k] + eps) + eps) : -logf(-logf(uu[k] + eps) + eps);
          yy[k] = (yy[k] + g) * temperature;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { y[e][k] = yy[k]; mx = fmaxf(mx, yy[k]); }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) y[e][k] = -INFINITY;
    }
  }
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  if (!orow) {
    // ids only (inference roll-outs): the softmax is monotone, so the first maximal index of y is the token -- no exp, no sum
#pragma unroll
    for (int e = 0; e < QPT; ++e) {
      const int q = tid + e * 1024;
      if (q < nq) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (y[e][k] > best) { best = y[e][k]; best_i = 4 * q + k; }
      }
    }
  } else {
    mx = block_max(mx, red);
    float sm = 0.f;
    float ex[QPT][4];
#pragma unroll
    for (int e = 0; e < QPT; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ex[e][k] = FAST ? __expf(y[e][k] - mx) : expf(y[e][k] - mx);       // exp(-inf) = 0 for the padding
        sm += ex[e][k];
      }
    sm = block_sum(sm, red);
    best = -1.f;
#pragma unroll
    for (int e = 0; e < QPT; ++e) {
      const int q = tid + e * 1024;
      if (q < nq) {
        float p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          p[k] = ex[e][k] / sm;
          if (p[k] > best) { best = p[k]; best_i = 4 * q + k; }
        }
        TA o4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o4[k] = from_f32<TA>(pretrain ? y[e][k] : p[k]);
        if constexpr (sizeof(TA) == 4) *(float4*)(orow + 4 * q) = *(const float4*)o4;
        else *(uint2*)(orow + 4 * q) = *(const uint2*)o4;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = best; red_i[tid >> 6] = best_i; }
  __syncthreads();
  best = red[0]; best_i = red_i[0];
  for (int w = 1; w < 16; ++w)
    if (red[w] > best || (red[w] == best && red_i[w] < best_i)) { best = red[w]; best_i = red_i[w]; }
  if (force_ids && (!force_len || t < force_len[b])) {
    const long f = force_ids[(long)b * ids_stride];
    best_i = (int)(f < 0 ? 0 : (f >= V ? V - 1 : f));
  }
  if (tid == 0) ids[(long)b * ids_stride] = best_i;
  if (x_next)
    for (int e = tid; e < E; e += 1024) x_next[(long)b * ld_x + e] = from_f32<TA>(embed[(long)best_i * E + e]);
}

// ---- softmax backward: dlogit = T * p * (dp - sum(dp*p))   (one block per (b,t) row)
template <typename TA>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const TA* __restrict__ p, const TA* __restrict__ dp,
                                                           TA* __restrict__ dl, float temperature_val, const float* __restrict__ t_dev, int V) {
  __shared__ float red[16];
  const float temperature = t_dev ? *t_dev : temperature_val;
  const long row = blockIdx.x;
  const TA* pr = p + row * V;
  const TA* dr = dp + row * V;
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) s += to_f32<TA>(pr[v]) * to_f32<TA>(dr[v]);
  s = block_sum(s, red);
  for (int v = threadIdx.x; v < V; v += 256) {
    const float pv = to_f32<TA>(pr[v]);
    dl[row * V + v] = from_f32<TA>(temperature * pv * (to_f32<TA>(dr[v]) - s));
  }
}

// ---- the same with the row held in registers: 16-byte loads of p and dp, all in flight at once, ONE pass over them (V % (16 / sizeof) == 0,
// V <= 256 * NCH * (16 / sizeof)); the scalar kernel above reads both rows twice, two bytes per lane at a time.
template <typename TA, int NCH>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const TA* __restrict__ p, const TA* __restrict__ dp,
                                                               TA* __restrict__ dl, float temperature_val, const float* __restrict__ t_dev, int V) {
  constexpr int VN = 16 / (int)sizeof(TA);
  __shared__ float red[16];
  const float temperature = t_dev ? *t_dev : temperature_val;
  const long row = blockIdx.x;
  const int nch = V / VN;
  TA pv[NCH][VN], dv[NCH][VN];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = threadIdx.x + 256 * i;
    const long o = row * V + (long)(c < nch ? c : 0) * VN;           // chunks past the row re-read chunk 0 (not used)
    *(uint4*)pv[i] = *(const uint4*)(p + o);
    *(uint4*)dv[i] = *(const uint4*)(dp + o);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (threadIdx.x + 256 * i < nch) {
#pragma unroll
      for (int k = 0; k < VN; ++k) s += to_f32<TA>(pv[i][k]) * to_f32<TA>(dv[i][k]);
    }
  }
  s = block_sum(s, red);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < nch) {
      TA ov[VN];
#pragma unroll
      for (int k = 0; k < VN; ++k) ov[k] = from_f32<TA>(temperature * to_f32<TA>(pv[i][k]) * (to_f32<TA>(dv[i][k]) - s));
      *(uint4*)(dl + row * V + (long)c * VN) = *(const uint4*)ov;
    }
  }
}

// Wcat = [w_ih | w_hh] in the compute dtype, bsum = b_ih + b_hh
template <typename TA>
__global__ void build_wcat_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                  const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                  TA* __restrict__ wcat, float* __restrict__ bsum, int rows, int din, int hid) {
  const long ld = din + hid;
  const long total = (long)rows * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ld;
    const int c = (int)(i % ld);
    const float v = c < din ? w_ih[r * din + c] : w_hh[r * hid + (c - din)];
    wcat[i] = from_f32<TA>(v);
    if (c == 0) bsum[r] = b_ih[r] + b_hh[r];
  }
}

// Resumed roll-outs: rows [r0, r1) join at this step with the recurrent state of caption r % srcB of an earlier call (one slot of
// its XH and c buffers): row copies of `rowbytes` in pieces of PB bytes (16 where the row size allows it, else 4 or 2: a bf16 row
// of (E + H) % 8 != 0 elements is not a whole number of 16-byte pieces) and H floats.
template <int PB>
__global__ void rollout_join_kernel(const unsigned char* __restrict__ src_xh, unsigned char* __restrict__ dst_xh, long rowbytes,
                                    const float* __restrict__ src_c, float* __restrict__ dst_c, int H, long r0, long r1, int srcB) {
  typedef typename std::conditional<PB == 16, uint4, typename std::conditional<PB == 4, uint32_t, uint16_t>::type>::type piece_t;
  const long pieces = rowbytes / PB;
  const long total = (r1 - r0) * pieces;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = r0 + i / pieces, p = i % pieces;
    *(piece_t*)(dst_xh + r * rowbytes + p * PB) = *(const piece_t*)(src_xh + (r % srcB) * rowbytes + p * PB);
  }
  const long totc = (r1 - r0) * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < totc; i += (long)gridDim.x * blockDim.x) {
    const long r = r0 + i / H;
    dst_c[r * H + i % H] = src_c[(r % srcB) * H + i % H];
  }
}

// ids-only generic roll-out step behind gemm_gumbelmax: token = index of the row's key (or the forced one), ids[b, t], and the token's
// embedding row into the next step's x slot; one 256-thread block per 4 rows
template <typename TA>
__global__ __launch_bounds__(256) void rollout_pick_kernel(const unsigned long long* __restrict__ rowkey, int64_t* __restrict__ ids, long ids_stride,
                                                           const float* __restrict__ embed, TA* __restrict__ x_next, long ld_x, int rows, int V,
                                                           int E, const int64_t* __restrict__ force_ids, const int32_t* __restrict__ force_len,
                                                           int t) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= rows) return;
  int id = row_key_index(rowkey[b]);
  if (force_ids && (!force_len || t < force_len[b])) id = (int)force_ids[(long)b * ids_stride];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  if (lane == 0) ids[(long)b * ids_stride] = id;
  if (x_next)
    for (int e = lane; e < E; e += 64) x_next[(long)b * ld_x + e] = from_f32<TA>(embed[(long)id * E + e]);
}

struct Ctx {
  int B, L, V, E, H, NL, dt;
  int din(int l) const { return l == 0 ? E : H; }
  long ldx(int l) const { return (long)din(l) + H; }
  size_t asz() const { return (size_t)dtype_size(dt); }
};

int check_dims(const gic_decoder_dims* d, Ctx& c) {
  GIC_CHECK_ARG(d, "decoder: null dims");
  GIC_CHECK_ARG(d->B > 0 && d->L > 0 && d->V > 1 && d->E > 0 && d->H > 0, "decoder: bad dims");
  GIC_CHECK_ARG(d->NL >= 1 && d->NL <= GIC_MAX_LAYERS, "decoder: gen_num_layers must be 1..%d", GIC_MAX_LAYERS);
  GIC_CHECK_ARG(d->dtype == DT_F32 || d->dtype == DT_BF16, "decoder: bad dtype");
  c = Ctx{d->B, d->L, d->V, d->E, d->H, d->NL, d->dtype};
  return GIC_OK;
}

// slot 0 of the recurrent buffers: initial (h, c) -- zeros, or the caller's states (generator.py:55,61) -- and features -> x_0
int init_slot0(const Ctx& c, const gic_decoder_state* st, const float* features, const gic_decoder_sample_opts* opt, hipStream_t stream) {
  const int B = c.B, E = c.E, H = c.H;
  for (int l = 0; l < c.NL; ++l) {
    if (opt && opt->h0) {
      if (l > 0) GIC_PROPAGATE(fill_zero(st->xh[l], (size_t)B * c.ldx(l) * c.asz(), stream));
      GIC_PROPAGATE(cast2d(opt->h0 + (long)l * B * H, DT_F32, H, (char*)st->xh[l] + (size_t)c.din(l) * c.asz(), c.dt, c.ldx(l), B, H, stream));
    } else {
      GIC_PROPAGATE(fill_zero(st->xh[l], (size_t)B * c.ldx(l) * c.asz(), stream));
    }
    if (opt && opt->c0) GIC_PROPAGATE(cast2d(opt->c0 + (long)l * B * H, DT_F32, H, st->c[l], DT_F32, H, B, H, stream));
    else GIC_PROPAGATE(fill_zero(st->c[l], (size_t)B * H * sizeof(float), stream));
  }
  return cast2d(features, DT_F32, E, st->xh[0], c.dt, c.ldx(0), B, E, stream);
}

// The roll-out as two fused launches per step + one finishing launch (decoder_step.h).
template <typename TA>
int sample_fwd_fused(const Ctx& c, const gic_decoder_params* P, const gic_decoder_shadow* S, const gic_decoder_state* st,
                     const float* noise_u, uint64_t seed, float temperature, int pretrain, void* out, int64_t* ids,
                     const gic_decoder_sample_opts* opt, hipStream_t stream) {
  const int B = c.B, L = c.L, V = c.V, H = c.H, NL = c.NL;
  const int nblk = cdiv(V, kVocabTile);
  const long per = (long)L * B * nblk;
  float* part_m = st->part;
  float* part_s = st->part + per;
  unsigned long long* rowkey = (unsigned long long*)(st->part + ((2 * per + 1) & ~1l));      // [L][B], 8-byte aligned
  // atomicMax targets start below every key
  GIC_PROPAGATE(fill_zero(rowkey, (size_t)L * B * sizeof(unsigned long long), stream));
  const bool keep = !(opt && opt->no_state);
  for (int t = 0; t < L; ++t) {
    for (int l = 0; l < NL; ++l) {
      const long ld = c.ldx(l);
      LstmStepArgs a;
      a.xh_t = (TA*)st->xh[l] + (long)t * B * ld;
      a.xh_next = (TA*)st->xh[l] + (long)(t + 1) * B * ld;
      a.wcat = S->wcat[l]; a.bsum = S->bsum[l];
      a.c_prev = st->c[l] + (long)t * B * H; a.c_new = st->c[l] + (long)(t + 1) * B * H;
      a.gates = keep ? st->gates[l] + (long)t * B * 4 * H : nullptr;
      if (l + 1 < NL) { a.h_up = (TA*)st->xh[l + 1] + (long)t * B * c.ldx(l + 1); a.ld_up = c.ldx(l + 1); }
      if (l + 1 == NL && st->hout) { a.h_out = (TA*)st->hout + (long)t * H; a.ld_out = (long)L * H; }
      a.B = B; a.H = H; a.din = c.din(l); a.ldx = ld;
      if (l == 0 && t > 0) {
        a.gather = 1; a.embed = P->embed; a.V = V;
        a.rowkey = rowkey + (long)(t - 1) * B;
        if (opt && opt->force_ids) { a.force_ids = opt->force_ids; a.force_stride = L; a.force_len = opt->force_len; }
        a.tprev = t - 1;
      }
      GIC_PROPAGATE(lstm_step(a, c.dt, stream));
    }
    VocabStepArgs v;
    const int l = NL - 1;
    v.h = (TA*)st->xh[l] + (long)(t + 1) * B * c.ldx(l) + c.din(l); v.ldh = c.ldx(l);
    v.wout = S->wout; v.bias = P->b_out;
    v.u = noise_u ? noise_u + (long)t * B * V : nullptr;
    v.seed = seed; v.rng_stream = (uint64_t)t; v.temperature = temperature; v.pretrain = pretrain;
    if (opt && opt->dev_scalars) { v.t_dev = &opt->dev_scalars->temperature; v.seed_dev = &opt->dev_scalars->seed[opt->seed_slot]; }
    v.out = out ? (void*)((TA*)out + (long)t * V) : nullptr; v.out_stride = (long)L * V;
    v.part_m = part_m + (long)t * B * nblk; v.part_s = part_s + (long)t * B * nblk; v.rowkey = rowkey + (long)t * B;
    v.nblk = nblk; v.B = B; v.V = V; v.H = H;
    GIC_PROPAGATE(vocab_step(v, c.dt, stream));
  }
  SampleFinishArgs f;
  f.part_m = part_m; f.part_s = part_s; f.rowkey = rowkey; f.nblk = nblk; f.B = B; f.L = L; f.V = V; f.E = c.E;
  f.pretrain = pretrain; f.out = out; f.ids = ids;
  if (opt && opt->force_ids) { f.force_ids = opt->force_ids; f.force_len = opt->force_len; }
  if (keep) { f.embed = P->embed; f.xh0 = st->xh[0]; f.ldx0 = c.ldx(0); }
  return sample_finish(f, c.dt, stream);
}

template <typename TA>
int sample_fwd_t(const Ctx& c, const gic_decoder_params* P, const gic_decoder_shadow* S, const gic_decoder_state* st,
                 const float* features, const float* noise_u, uint64_t seed, float temperature, int pretrain,
                 void* out, int64_t* ids, const gic_decoder_sample_opts* opt, hipStream_t stream) {
  const int B = c.B, L = c.L, V = c.V, E = c.E, H = c.H, NL = c.NL;
  const int pw_grid = cdiv((long)B * H, 256);
  GIC_PROPAGATE(init_slot0(c, st, features, opt, stream));
  // up to a few hundred rows the per-step products are latency-bound: the fused step kernels; beyond that (Monte-Carlo
  // roll-out batches) they are large GEMMs and the generic 128-row-tile kernels are the efficient form
  if (st->part && B <= decoder_step_max_rows() && !(opt && opt->resume_from) && decoder_step_supported(c.dt, V, E, H, NL))
    return sample_fwd_fused<TA>(c, P, S, st, noise_u, seed, temperature, pretrain, out, ids, opt, stream);
  if (opt && opt->dev_scalars) {
    set_last_error("decoder_sample_fwd: device-resident step scalars need the fused step kernels (state->part, B <= %d, V %% 4 == 0, E, H %% 8 == 0)",
                   decoder_step_max_rows());
    return GIC_ERR_UNSUPPORTED;
  }
  GIC_CHECK_ARG(st->logits && st->gpre, "decoder_sample_fwd: the unfused path needs state->logits and state->gpre");
  const bool keep = !(opt && opt->no_state);
  const int64_t* f_ids = opt ? opt->force_ids : nullptr;
  const int32_t* f_len = opt ? opt->force_len : nullptr;
  // resumed roll-outs: at step t only the first act[t] rows exist; the rows that join take their state from the earlier call
  const int32_t* act = (opt && opt->resume_from) ? opt->host_active_rows : nullptr;
  if (act) GIC_PROPAGATE(hipMemcpyAsync(ids, f_ids, (size_t)B * L * sizeof(int64_t), hipMemcpyDeviceToDevice, stream) == hipSuccess ? GIC_OK : GIC_ERR_LAUNCH);
  // ids-only roll-outs in the bf16 mode: one 64-bit argmax key per (step, row) in the logits scratch (gemm_gumbelmax); zeroed once
  unsigned long long* rowkeys = nullptr;
  if (!out && !pretrain && c.dt == DT_BF16 && V % 4 == 0 && (long)L * 8 <= (long)V * 4 && ((uintptr_t)st->logits & 7) == 0) {
    rowkeys = (unsigned long long*)st->logits;                  // [L][B] keys inside the [B, V] f32 scratch (L * 8 <= V * 4 bytes per row)
    GIC_PROPAGATE(fill_zero(rowkeys, (size_t)L * B * sizeof(unsigned long long), stream));
  }

  for (int t = 0; t < L; ++t) {
    const int M = act ? act[t] : B;                     // rows that take part in this step
    if (act) {
      const int M_prev = t > 0 ? act[t - 1] : 0;
      if (M > M_prev) {
        for (int l = 0; l < NL; ++l) {
          const long rowbytes = c.ldx(l) * (long)c.asz();
          const int pb = rowbytes % 16 == 0 ? 16 : (rowbytes % 4 == 0 ? 4 : 2);      // rowbytes is a multiple of the element size
          const long work = (long)(M - M_prev) * (rowbytes / pb);
          const dim3 jgrid((unsigned)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256));
          const unsigned char* jsrc = (const unsigned char*)opt->resume_from->xh[l] + (size_t)t * opt->resume_B * rowbytes;
          unsigned char* jdst = (unsigned char*)st->xh[l] + (size_t)t * B * rowbytes;
          const float* jsc = (const float*)(opt->resume_from->c[l] + (long)t * opt->resume_B * H);
          float* jdc = st->c[l] + (long)t * B * H;
          if (pb == 16) hipLaunchKernelGGL(rollout_join_kernel<16>, jgrid, dim3(256), 0, stream, jsrc, jdst, rowbytes, jsc, jdc, H, (long)M_prev, (long)M, opt->resume_B);
          else if (pb == 4) hipLaunchKernelGGL(rollout_join_kernel<4>, jgrid, dim3(256), 0, stream, jsrc, jdst, rowbytes, jsc, jdc, H, (long)M_prev, (long)M, opt->resume_B);
          else hipLaunchKernelGGL(rollout_join_kernel<2>, jgrid, dim3(256), 0, stream, jsrc, jdst, rowbytes, jsc, jdc, H, (long)M_prev, (long)M, opt->resume_B);
          GIC_CHECK_LAUNCH("rollout_join");
        }
      }
      if (M == 0) continue;
    }
    const int pw_grid_t = cdiv((long)M * H, 256);
    for (int l = 0; l < NL; ++l) {
      const long ld = c.ldx(l);
      TA* xh_t = (TA*)st->xh[l] + (long)t * B * ld;
      TA* xh_n = (TA*)st->xh[l] + (long)(t + 1) * B * ld;
      GemmDesc g;
      g.A = xh_t; g.lda = ld; g.B = S->wcat[l]; g.ldb = ld; g.C = st->gpre; g.ldc = 4 * H;
      g.M = M; g.N = 4 * H; g.K = (int)ld; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.bias = S->bsum[l];
      GIC_PROPAGATE(gemm(g, stream));
      TA* h_up = (l + 1 < NL) ? (TA*)st->xh[l + 1] + (long)t * B * c.ldx(l + 1) : nullptr;
      TA* h_out = (l + 1 == NL && st->hout) ? (TA*)st->hout + (long)t * H : nullptr;
      hipLaunchKernelGGL((lstm_pointwise_fwd_kernel<TA>), dim3(pw_grid_t), dim3(256), 0, stream,
                         (const float*)st->gpre, (const float*)(st->c[l] + (long)t * B * H),
                         keep ? st->gates[l] + (long)t * B * 4 * H : (float*)nullptr, st->c[l] + (long)(t + 1) * B * H,
                         xh_n + c.din(l), ld, h_up, h_up ? c.ldx(l + 1) : 0, h_out, (long)L * H, M, H);
      GIC_CHECK_LAUNCH("lstm_pointwise_fwd");
    }
    TA* x_next = (TA*)st->xh[0] + (long)(t + 1) * B * c.ldx(0);
    const float* u_t = noise_u ? noise_u + (long)t * B * V : nullptr;
    if (!out && !pretrain && rowkeys) {
      // ids only (Monte-Carlo roll-outs): vocabulary product with the Gumbel-max in its epilogue -- the [rows, V] logits never reach
      // memory (389 MB written and read again per step at 9728 rows), then key -> token -> next input row
      const int l = NL - 1;
      const long ld = c.ldx(l);
      GemmDesc g;
      g.A = S->wout; g.lda = H; g.B = (TA*)st->xh[l] + (long)(t + 1) * B * ld + c.din(l); g.ldb = ld;
      g.M = V; g.N = M; g.K = H; g.in_dtype = c.dt; g.out_dtype = DT_F32;
      g.gm_rowkey = rowkeys + (long)t * B; g.gm_bias = P->b_out; g.gm_u = u_t; g.gm_ldu = V; g.gm_temperature = temperature;
      g.seed = seed; g.stream = (uint64_t)t;
      const int s = gemm_gumbelmax(g, stream);
      if (s == GIC_OK) {
        hipLaunchKernelGGL((rollout_pick_kernel<TA>), dim3((unsigned)cdiv(M, 4)), dim3(256), 0, stream, (const unsigned long long*)g.gm_rowkey,
                           ids + t, (long)L, P->embed, x_next, c.ldx(0), M, V, E, f_ids ? f_ids + t : nullptr, f_len, t);
        GIC_CHECK_LAUNCH("rollout_pick");
        continue;
      }
      if (s != GIC_ERR_UNSUPPORTED) return s;
      rowkeys = nullptr;                        // shapes the fused product declines: the separate launches from here on
    }
    {  // vocabulary projection on the last layer's h_t (lives in XH_last[t+1][:, Din:])
      const int l = NL - 1;
      const long ld = c.ldx(l);
      GemmDesc g;
      g.A = (TA*)st->xh[l] + (long)(t + 1) * B * ld + c.din(l); g.lda = ld;
      g.B = S->wout; g.ldb = H; g.C = st->logits; g.ldc = V;
      g.M = M; g.N = V; g.K = H; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.bias = P->b_out;
      GIC_PROPAGATE(gemm(g, stream));
    }
    TA* out_t = out ? (TA*)out + (long)t * V : nullptr;
    constexpr bool kFast = sizeof(TA) == 2;
    if (V % 4 == 0 && V <= 4096) {
      hipLaunchKernelGGL((gumbel_softmax_argmax_reg_kernel<TA, 1, kFast>), dim3(M), dim3(1024), 0, stream, (const float*)st->logits,
                         u_t, seed, (uint64_t)t, temperature, pretrain, out_t, (long)L * V, ids + t, (long)L,
                         P->embed, x_next, c.ldx(0), V, E, f_ids ? f_ids + t : nullptr, f_len, t);
    } else if (V % 4 == 0 && V <= 16384) {
      hipLaunchKernelGGL((gumbel_softmax_argmax_reg_kernel<TA, 4, kFast>), dim3(M), dim3(1024), 0, stream, (const float*)st->logits,
                         u_t, seed, (uint64_t)t, temperature, pretrain, out_t, (long)L * V, ids + t, (long)L,
                         P->embed, x_next, c.ldx(0), V, E, f_ids ? f_ids + t : nullptr, f_len, t);
    } else {
      hipLaunchKernelGGL((gumbel_softmax_argmax_kernel<TA>), dim3(M), dim3(256), 0, stream, st->logits, u_t, seed, (uint64_t)t,
                         temperature, pretrain, out_t, (long)L * V, ids + t, (long)L, P->embed, x_next,
                         c.ldx(0), V, E, f_ids ? f_ids + t : nullptr, f_len, t);
    }
    GIC_CHECK_LAUNCH("gumbel_softmax_argmax");
  }
  return GIC_OK;
}

template <typename TA>
int sample_bwd_t(const Ctx& c, const gic_decoder_params* P, const gic_decoder_shadow* S, const gic_decoder_state* st,
                 const gic_decoder_bwd_ws* ws, const void* probs, const int64_t* ids, const void* d_out,
                 float temperature, const float* t_dev, int pretrain, const gic_decoder_grads* G, int phases, hipStream_t stream) {
  const int B = c.B, L = c.L, V = c.V, E = c.E, H = c.H, NL = c.NL;
  const long BL = (long)B * L;
  const int pw_grid = cdiv((long)B * H, 256);
  // 1. + 2. output layer: d_logits [B,L,V], d_hout = d_logits W_out, dW_out = d_logits^T hout, db_out = colsum(d_logits)
  if (phases & GIC_DECODER_BWD_OUTPUT)
    GIC_PROPAGATE(decoder_output_bwd(c.dt, B, L, V, H, probs, d_out, temperature, t_dev, pretrain, ws->dlogits, S->wout, st->hout, ws->dhout,
                                     G->w_out, G->b_out, stream));
  if (!(phases & GIC_DECODER_BWD_RECURRENT)) return GIC_OK;
  // 3. BPTT
  bool fused = decoder_step_supported(c.dt, 4, 8, H, NL) && H % 2 == 0 && B <= decoder_step_max_rows();
  for (int l = 0; l < NL; ++l) fused = fused && S->wcat_t[l] != nullptr;
  for (int l = 0; l < NL; ++l) GIC_PROPAGATE(fill_zero(ws->dc[l], (size_t)B * H * sizeof(float), stream));
  if (fused) {
    // one launch per (step, layer): the recurrent (and, below the top layer, the upper layer's input) gradient product fused with
    // the cell's pointwise backward (decoder_step.h); the x-side input gradients leave the serial chain as ONE product afterwards
    for (int t = L - 1; t >= 0; --t) {
      for (int l = NL - 1; l >= 0; --l) {
        LstmBwdStepArgs a;
        if (l == NL - 1) { a.dh_above = ws->dhout + (long)t * H; a.ld_above = (long)L * H; }
        else { a.dg_up = (TA*)ws->dgates[l + 1] + (long)t * B * 4 * H; a.w_up = S->wcat_t[l + 1]; }
        if (t + 1 < L) { a.dg_next = (TA*)ws->dgates[l] + (long)(t + 1) * B * 4 * H; a.w_rec = (TA*)S->wcat_t[l] + (long)c.din(l) * 4 * H; }
        a.gates = st->gates[l] + (long)t * B * 4 * H;
        a.c_prev = st->c[l] + (long)t * B * H; a.c_cur = st->c[l] + (long)(t + 1) * B * H;
        a.dc_state = ws->dc[l]; a.dgates = (TA*)ws->dgates[l] + (long)t * B * 4 * H;
        a.B = B; a.H = H;
        GIC_PROPAGATE(lstm_bwd_step(a, c.dt, stream));
      }
    }
    {  // d x_t of layer 0 for every step: dX0 [L*B, E] = dgates_0 [L*B, 4H] . W_ih_0, into the x columns of dxh[0] slots 0..L-1
      GemmDesc g;
      g.A = ws->dgates[0]; g.lda = 4 * H; g.a_kc = 1; g.B = S->wcat_t[0]; g.ldb = 4 * H; g.b_kc = 1;
      g.C = ws->dxh[0]; g.ldc = c.ldx(0); g.M = (int)BL; g.N = E; g.K = 4 * H; g.in_dtype = c.dt; g.out_dtype = DT_F32;
      GIC_PROPAGATE(gemm(g, stream));
    }
    if (phases & GIC_DECODER_BWD_STATE_GRADS) {   // d h_-1 = dgates_0 . W_hh into the h columns of slot 0 (gradient of the initial state)
      for (int l = 0; l < NL; ++l) {
        GemmDesc g;
        g.A = ws->dgates[l]; g.lda = 4 * H; g.a_kc = 1; g.B = (TA*)S->wcat_t[l] + (long)c.din(l) * 4 * H; g.ldb = 4 * H; g.b_kc = 1;
        g.C = ws->dxh[l] + c.din(l); g.ldc = c.ldx(l); g.M = B; g.N = H; g.K = 4 * H; g.in_dtype = c.dt; g.out_dtype = DT_F32;
        GIC_PROPAGATE(gemm(g, stream));
      }
    }
  } else {
  for (int l = 0; l < NL; ++l) {
    // all L+1 slots at once: slot L is the zero gradient behind the last step, slots < L are split-K accumulators
    GIC_PROPAGATE(fill_zero(ws->dxh[l], (size_t)(L + 1) * B * c.ldx(l) * sizeof(float), stream));
  }
  for (int t = L - 1; t >= 0; --t) {
    for (int l = NL - 1; l >= 0; --l) {
      const long ld = c.ldx(l);
      const float* dh_a;
      long ld_a;
      if (l == NL - 1) { dh_a = ws->dhout + (long)t * H; ld_a = (long)L * H; }
      else { dh_a = ws->dxh[l + 1] + (long)t * B * c.ldx(l + 1); ld_a = c.ldx(l + 1); }
      const float* dh_b = ws->dxh[l] + (long)(t + 1) * B * ld + c.din(l);
      TA* dg = (TA*)ws->dgates[l] + (long)t * B * 4 * H;
      hipLaunchKernelGGL((lstm_pointwise_bwd_kernel<TA>), dim3(pw_grid), dim3(256), 0, stream, dh_a, ld_a, dh_b, ld,
                         (const float*)(st->gates[l] + (long)t * B * 4 * H), (const float*)(st->c[l] + (long)t * B * H),
                         (const float*)(st->c[l] + (long)(t + 1) * B * H), ws->dc[l], dg, B, H);
      GIC_CHECK_LAUNCH("lstm_pointwise_bwd");
      GemmDesc g;   // d[x | h_prev] = d_gates Wcat
      g.A = dg; g.lda = 4 * H; g.a_kc = 1;
      if (S->wcat_t[l]) { g.B = S->wcat_t[l]; g.ldb = 4 * H; g.b_kc = 1; }
      else { g.B = S->wcat[l]; g.ldb = ld; g.b_kc = 0; }
      g.C = ws->dxh[l] + (long)t * B * ld; g.ldc = ld; g.c_zeroed = 1;
      g.M = B; g.N = (int)ld; g.K = 4 * H; g.in_dtype = c.dt; g.out_dtype = DT_F32;
      GIC_PROPAGATE(gemm(g, stream));
    }
  }
  }
  // 4. batched weight gradients over all L*B rows
  for (int l = 0; l < NL; ++l) {
    const long ld = c.ldx(l);
    GemmDesc w;
    w.A = ws->dgates[l]; w.lda = 4 * H; w.a_kc = 0; w.b_kc = 0; w.ldb = ld;
    w.M = 4 * H; w.K = (int)BL; w.in_dtype = c.dt; w.out_dtype = DT_F32;
    w.B = st->xh[l]; w.N = c.din(l); w.C = G->w_ih[l]; w.ldc = c.din(l);
    GIC_PROPAGATE(gemm(w, stream));
    w.B = (const TA*)st->xh[l] + c.din(l); w.N = H; w.C = G->w_hh[l]; w.ldc = H;
    GIC_PROPAGATE(gemm(w, stream));
    GIC_PROPAGATE(colsum(ws->dgates[l], c.dt, 4 * H, BL, 4 * H, G->b_ih[l], G->b_hh[l], 0, stream));
  }
  // 5. inputs: d_features = dx_0 ; d_embed[ids[b,t-1]] += dx_t (t >= 1)  (generator.py:75: index detached)
  GIC_PROPAGATE(cast2d(ws->dxh[0], DT_F32, c.ldx(0), G->features, DT_F32, E, B, E, stream));
  // the embedding scatter-add (embed_scatter_time) is enqueued by the caller below
  return GIC_OK;
}

__global__ void embed_scatter_time_kernel(const float* __restrict__ dxh0, long ld, const int64_t* __restrict__ ids, long ids_stride,
                                          float* __restrict__ dw, int B, int L, int E, int V) {
  const long total = (long)(L - 1) * B * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    const long r = i / E;
    const int b = (int)(r % B);
    const int t = (int)(r / B) + 1;
    long id = ids[(long)b * ids_stride + (t - 1)];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    atomicAdd(&dw[id * E + e], dxh0[((long)t * B + b) * ld + e]);
  }
}

}  // namespace

// Output layer of the decoder, backward (shared by the LSTM and the attention decoder): softmax/Gumbel backward (adversarial mode),
// d_hout, and the complete gradients of the vocabulary projection.
int decoder_output_bwd(int dt, int B, int L, int V, int H, const void* probs, const void* d_out, float temperature, const float* t_dev, int pretrain,
                       void* dlogits_ws, const void* wout, const void* hout, float* dhout, float* d_wout, float* d_bout, hipStream_t stream) {
  const long BL = (long)B * L;
  const void* dlog = pretrain ? d_out : (const void*)dlogits_ws;
  if (!pretrain) {
    const bool al = ((((uintptr_t)probs) | ((uintptr_t)d_out) | ((uintptr_t)dlogits_ws)) & 15) == 0;
    if (dt == DT_F32) {
      if (al && V % 4 == 0 && V <= 256 * 8 * 4)
        hipLaunchKernelGGL((softmax_bwd_vec_kernel<float, 8>), dim3((unsigned)BL), dim3(256), 0, stream, (const float*)probs, (const float*)d_out,
                           (float*)dlogits_ws, temperature, t_dev, V);
      else
        hipLaunchKernelGGL((softmax_bwd_kernel<float>), dim3((unsigned)BL), dim3(256), 0, stream, (const float*)probs, (const float*)d_out,
                           (float*)dlogits_ws, temperature, t_dev, V);
    } else {
      if (al && V % 8 == 0 && V <= 256 * 5 * 8)
        hipLaunchKernelGGL((softmax_bwd_vec_kernel<bf16_t, 5>), dim3((unsigned)BL), dim3(256), 0, stream, (const bf16_t*)probs, (const bf16_t*)d_out,
                           (bf16_t*)dlogits_ws, temperature, t_dev, V);
      else if (al && V % 8 == 0 && V <= 256 * 8 * 8)
        hipLaunchKernelGGL((softmax_bwd_vec_kernel<bf16_t, 8>), dim3((unsigned)BL), dim3(256), 0, stream, (const bf16_t*)probs, (const bf16_t*)d_out,
                           (bf16_t*)dlogits_ws, temperature, t_dev, V);
      else
        hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), dim3((unsigned)BL), dim3(256), 0, stream, (const bf16_t*)probs, (const bf16_t*)d_out,
                           (bf16_t*)dlogits_ws, temperature, t_dev, V);
    }
    GIC_CHECK_LAUNCH("softmax_bwd");
  }
  GemmDesc g;
  g.A = dlog; g.lda = V; g.a_kc = 1; g.B = wout; g.ldb = H; g.b_kc = 0; g.C = dhout; g.ldc = H;
  g.M = (int)BL; g.N = H; g.K = V; g.in_dtype = dt; g.out_dtype = DT_F32;
  GIC_PROPAGATE(gemm(g, stream));
  GemmDesc w;
  w.A = dlog; w.lda = V; w.a_kc = 0; w.B = hout; w.ldb = H; w.b_kc = 0; w.C = d_wout; w.ldc = H;
  w.M = V; w.N = H; w.K = (int)BL; w.in_dtype = dt; w.out_dtype = DT_F32;
  GIC_PROPAGATE(gemm(w, stream));
  return colsum(dlog, dt, V, BL, V, d_bout, nullptr, 0, stream);
}

// d_embed[ids[b, t-1]] += dx_t (t >= 1) over a zeroed table (generator.py:75: the index is detached); dx rows at dx + (t*B + b)*ld,
// ids rows ids_stride apart (0: L)
int embed_scatter_time(const float* dx, long ld, const int64_t* ids, float* d_embed, int B, int L, int E, int V, hipStream_t stream, long ids_stride) {
  if (ids_stride <= 0) ids_stride = L;
  GIC_PROPAGATE(fill_zero(d_embed, (size_t)V * E * sizeof(float), stream));
  if (L > 1) {
    const long total = (long)(L - 1) * B * E;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(embed_scatter_time_kernel, dim3(grid), dim3(256), 0, stream, dx, ld, ids, ids_stride, d_embed, B, L, E, V);
    GIC_CHECK_LAUNCH("embed_scatter_time");
  }
  return GIC_OK;
}

}  // namespace gic

using namespace gic;

// ---- teacher-forced decode (Decoder.forward, generator.py:39-53): inputs known up front, packed-sequence semantics
// x rows of slots 1..T-1 of XH_0: embed(caps[b, t-1])
template <typename TA>
__global__ void embed_rows_tf_kernel(const float* __restrict__ embed, const int64_t* __restrict__ caps, TA* __restrict__ xh0, long ld,
                                     int B, int Tm1, int E, int V) {
  const long total = (long)Tm1 * B * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    const long r = i / E;
    const int b = (int)(r % B), t = (int)(r / B) + 1;
    long id = caps[(long)b * Tm1 + (t - 1)];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    xh0[((long)t * B + b) * ld + e] = from_f32<TA>(embed[id * E + e]);
  }
}

// LSTM pointwise with pack_padded_sequence semantics: a row past its length keeps (h, c) and outputs zero
template <typename TA>
__global__ void lstm_pointwise_tf_kernel(const float* __restrict__ gpre, const float* __restrict__ c_prev, const TA* __restrict__ h_prev,
                                         long ld_prev, float* __restrict__ c_new, TA* __restrict__ h_next, long ld_next,
                                         TA* __restrict__ h_up, long ld_up, TA* __restrict__ h_out, long ld_out,
                                         const int32_t* __restrict__ lengths, int t, int B, int H, float* __restrict__ gates_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  float* go = gates_out ? gates_out + (long)b * 4 * H + j : nullptr;   // saved for gic_decoder_forward_tf_bwd, laid out as lstm_step saves them
  if (t >= lengths[b]) {
    if (go) { go[0] = 0.f; go[H] = 0.f; go[2 * H] = 0.f; go[3 * H] = 0.f; }   // never weighted by a non-zero gradient: finite is all that matters
    c_new[idx] = c_prev[idx];
    const TA hp = h_prev[(long)b * ld_prev + j];
    h_next[(long)b * ld_next + j] = hp;
    if (h_up) h_up[(long)b * ld_up + j] = hp;
    if (h_out) h_out[(long)b * ld_out + j] = from_f32<TA>(0.f);
    return;
  }
  const float* g = gpre + (long)b * 4 * H;
  const float i_ = sigmoidf_(g[j]);
  const float f_ = sigmoidf_(g[H + j]);
  const float g_ = tanhf(g[2 * H + j]);
  const float o_ = sigmoidf_(g[3 * H + j]);
  const float c = f_ * c_prev[idx] + i_ * g_;
  const float h = o_ * tanhf(c);
  if (go) { go[0] = i_; go[H] = f_; go[2 * H] = g_; go[3 * H] = o_; }
  c_new[idx] = c;
  h_next[(long)b * ld_next + j] = from_f32<TA>(h);
  if (h_up) h_up[(long)b * ld_up + j] = from_f32<TA>(h);
  if (h_out) h_out[(long)b * ld_out + j] = from_f32<TA>(h);
}

// d_hout [B, Tmax, H]: rows past their sequence's length are pad_packed_sequence's constant zeros (generator.py:45): no gradient reaches the LSTM
__global__ void zero_past_length_kernel(float* __restrict__ dhout, const int32_t* __restrict__ lengths, int B, int Tmax, int H) {
  const long total = (long)B * Tmax * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / H;
    const int b = (int)(r / Tmax), t = (int)(r % Tmax);
    if (t >= lengths[b]) dhout[i] = 0.f;
  }
}

template <typename TA>
int forward_tf_t(const Ctx& c, const gic_decoder_params* P, const gic_decoder_shadow* S, const gic_decoder_state* st,
                 const float* features, const int64_t* caps, const int32_t* lengths, int Tmax, const float* noise_u, uint64_t seed,
                 float temperature, int pretrain, float* logits_ws, int64_t* ids_ws, void* out, float* h_n, float* c_n,
                 hipStream_t stream) {
  const int B = c.B, T = c.L, V = c.V, E = c.E, H = c.H, NL = c.NL;
  const int pw_grid = cdiv((long)B * H, 256);
  for (int l = 0; l < NL; ++l) {
    GIC_PROPAGATE(fill_zero(st->xh[l], (size_t)B * c.ldx(l) * c.asz(), stream));
    GIC_PROPAGATE(fill_zero(st->c[l], (size_t)B * H * sizeof(float), stream));
  }
  GIC_PROPAGATE(cast2d(features, DT_F32, E, st->xh[0], c.dt, c.ldx(0), B, E, stream));
  if (T > 1) {
    const long total = (long)(T - 1) * B * E;
    hipLaunchKernelGGL((embed_rows_tf_kernel<TA>), dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                       stream, P->embed, caps, (TA*)st->xh[0], c.ldx(0), B, T - 1, E, V);
    GIC_CHECK_LAUNCH("embed_rows_tf");
  }
  for (int t = 0; t < Tmax; ++t) {
    for (int l = 0; l < NL; ++l) {
      const long ld = c.ldx(l);
      TA* xh_t = (TA*)st->xh[l] + (long)t * B * ld;
      TA* xh_n = (TA*)st->xh[l] + (long)(t + 1) * B * ld;
      GemmDesc g;
      g.A = xh_t; g.lda = ld; g.B = S->wcat[l]; g.ldb = ld; g.C = st->gpre; g.ldc = 4 * H;
      g.M = B; g.N = 4 * H; g.K = (int)ld; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.bias = S->bsum[l];
      GIC_PROPAGATE(gemm(g, stream));
      TA* h_up = (l + 1 < NL) ? (TA*)st->xh[l + 1] + (long)t * B * c.ldx(l + 1) : nullptr;
      TA* h_out = (l + 1 == NL) ? (TA*)st->hout + (long)t * H : nullptr;            // hout viewed as [B, Tmax, H]
      hipLaunchKernelGGL((lstm_pointwise_tf_kernel<TA>), dim3(pw_grid), dim3(256), 0, stream, (const float*)st->gpre,
                         (const float*)(st->c[l] + (long)t * B * H), (const TA*)(xh_t + c.din(l)), ld,
                         st->c[l] + (long)(t + 1) * B * H, xh_n + c.din(l), ld, h_up, h_up ? c.ldx(l + 1) : 0, h_out, (long)Tmax * H,
                         lengths, t, B, H, st->gates[l] ? st->gates[l] + (long)t * B * 4 * H : nullptr);
      GIC_CHECK_LAUNCH("lstm_pointwise_tf");
    }
  }
  // one projection over all B*Tmax rows, then (adversarial mode) Gumbel + softmax per row; the draw u is [B, Tmax, V]
  const long rows = (long)B * Tmax;
  {
    GemmDesc g;
    g.A = st->hout; g.lda = H; g.B = S->wout; g.ldb = H; g.C = logits_ws; g.ldc = V;
    g.M = (int)rows; g.N = V; g.K = H; g.in_dtype = c.dt; g.out_dtype = DT_F32; g.bias = P->b_out;
    GIC_PROPAGATE(gemm(g, stream));
  }
  hipLaunchKernelGGL((gumbel_softmax_argmax_kernel<TA>), dim3((unsigned)rows), dim3(256), 0, stream, logits_ws, noise_u, seed, (uint64_t)0x7466,
                     temperature, pretrain, (TA*)out, (long)V, ids_ws, (long)1, P->embed, (TA*)nullptr, (long)0, V, E,
                     (const int64_t*)nullptr, (const int32_t*)nullptr, 0);
  GIC_CHECK_LAUNCH("gumbel_softmax (teacher forced)");
  for (int l = 0; l < NL; ++l) {
    GIC_PROPAGATE(cast2d((const TA*)st->xh[l] + (long)Tmax * B * c.ldx(l) + c.din(l), c.dt, c.ldx(l), h_n + (long)l * B * H, DT_F32, H, B, H, stream));
    GIC_PROPAGATE(cast2d(st->c[l] + (long)Tmax * B * H, DT_F32, H, c_n + (long)l * B * H, DT_F32, H, B, H, stream));
  }
  return GIC_OK;
}

extern "C" {

int gic_decoder_forward_tf(const gic_decoder_dims* dims, const gic_decoder_params* P, const gic_decoder_shadow* S,
                           const gic_decoder_state* st, const float* features, const int64_t* caps, const int32_t* lengths, int Tmax,
                           const float* noise_u, uint64_t seed, float temperature, int pretrain, float* logits_ws, int64_t* ids_ws,
                           void* out, float* h_n, float* c_n, void* stream) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(P && S && st && features && lengths && logits_ws && ids_ws && out && h_n && c_n, "decoder_forward_tf: null argument");
  GIC_CHECK_ARG(c.L == 1 || caps, "decoder_forward_tf: caps is null");
  GIC_CHECK_ARG(Tmax >= 1 && Tmax <= c.L, "decoder_forward_tf: Tmax must be in 1..L (= caption length + 1)");
  GIC_CHECK_ARG(P->embed && P->b_out && S->wout && st->hout && st->gpre, "decoder_forward_tf: null buffer");
  for (int l = 0; l < c.NL; ++l)
    GIC_CHECK_ARG(st->xh[l] && st->c[l] && S->wcat[l] && S->bsum[l], "decoder_forward_tf: null layer %d buffer", l);
  if (c.dt == DT_F32)
    return forward_tf_t<float>(c, P, S, st, features, caps, lengths, Tmax, noise_u, seed, temperature, pretrain, logits_ws, ids_ws, out,
                               h_n, c_n, (hipStream_t)stream);
  return forward_tf_t<bf16_t>(c, P, S, st, features, caps, lengths, Tmax, noise_u, seed, temperature, pretrain, logits_ws, ids_ws, out,
                              h_n, c_n, (hipStream_t)stream);
}

int gic_decoder_forward_tf_bwd(const gic_decoder_dims* dims, const gic_decoder_params* P, const gic_decoder_shadow* S,
                               const gic_decoder_state* st, const gic_decoder_bwd_ws* ws, const void* pred, const int64_t* caps,
                               const int32_t* lengths, int Tmax, const void* d_pred, float temperature, int pretrain,
                               const gic_decoder_grads* G, void* stream_) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(P && S && st && ws && pred && lengths && d_pred && G, "decoder_forward_tf_bwd: null argument");
  GIC_CHECK_ARG(c.L == 1 || caps, "decoder_forward_tf_bwd: caps is null");
  GIC_CHECK_ARG(Tmax >= 1 && Tmax <= c.L, "decoder_forward_tf_bwd: Tmax must be in 1..L (= caption length + 1)");
  GIC_CHECK_ARG(ws->dlogits && ws->dhout && st->hout && S->wout && G->embed && G->w_out && G->b_out && G->features, "decoder_forward_tf_bwd: null buffer");
  for (int l = 0; l < c.NL; ++l)
    GIC_CHECK_ARG(st->xh[l] && st->c[l] && st->gates[l] && ws->dgates[l] && ws->dxh[l] && ws->dc[l] && G->w_ih[l] && G->w_hh[l] && G->b_ih[l] && G->b_hh[l],
                  "decoder_forward_tf_bwd: null layer %d buffer (the forward call must have been given state->gates)", l);
  hipStream_t stream = (hipStream_t)stream_;
  const int T = c.L;
  // the saved state and every [B, Tmax, .] tensor are laid out for Tmax steps: run the sampled path's backward on that view
  c.L = Tmax;
  GIC_PROPAGATE(decoder_output_bwd(c.dt, c.B, Tmax, c.V, c.H, pred, d_pred, temperature, nullptr, pretrain, ws->dlogits, S->wout, st->hout, ws->dhout,
                                   G->w_out, G->b_out, stream));
  {
    const long total = (long)c.B * Tmax * c.H;
    hipLaunchKernelGGL(zero_past_length_kernel, dim3((unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0, stream,
                       ws->dhout, lengths, c.B, Tmax, c.H);
    GIC_CHECK_LAUNCH("zero_past_length");
  }
  // caps stands in for the sampled ids (the input of step t is embed(caps[b, t-1])); sample_bwd_t itself never reads them
  const int s = (c.dt == DT_F32)
                    ? sample_bwd_t<float>(c, P, S, st, ws, pred, caps, d_pred, temperature, nullptr, pretrain, G, GIC_DECODER_BWD_RECURRENT, stream)
                    : sample_bwd_t<bf16_t>(c, P, S, st, ws, pred, caps, d_pred, temperature, nullptr, pretrain, G, GIC_DECODER_BWD_RECURRENT, stream);
  GIC_PROPAGATE(s);
  return embed_scatter_time((const float*)ws->dxh[0], c.ldx(0), caps, G->embed, c.B, Tmax, c.E, c.V, stream, T - 1);
}

void gic_debug_decoder_step(int v) { decoder_step_debug(v); }

// one thread: *dev = v (gic_step_scalars_set; the values travel as a kernel argument)
__global__ void step_scalars_set_kernel(gic_step_scalars* dev, const gic_step_scalars v) { *dev = v; }

int gic_step_scalars_set(gic_step_scalars* dev, const gic_step_scalars* host_values, void* stream) {
  GIC_CHECK_ARG(dev && host_values, "step_scalars_set: null argument");
  hipLaunchKernelGGL(step_scalars_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dev, *host_values);
  GIC_CHECK_LAUNCH("step_scalars_set");
  return GIC_OK;
}

int gic_decoder_fused_rollout_rows(const gic_decoder_dims* dims, int32_t* out) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(out, "decoder_fused_rollout_rows: null out");
  *out = decoder_step_supported(c.dt, c.V, c.E, c.H, c.NL) ? decoder_step_max_rows() : 0;
  return GIC_OK;
}

int gic_decoder_state_bytes(const gic_decoder_dims* dims, uint64_t* out) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(out, "decoder_state_bytes: null out");
  const uint64_t B = c.B, L = c.L, H = c.H, a = c.asz();
  for (int l = 0; l < GIC_MAX_LAYERS; ++l) {
    const bool on = l < c.NL;
    out[l] = on ? (L + 1) * B * (uint64_t)c.ldx(l) * a : 0;
    out[GIC_MAX_LAYERS + l] = on ? L * B * 4 * H * 4 : 0;
    out[2 * GIC_MAX_LAYERS + l] = on ? (L + 1) * B * H * 4 : 0;
  }
  out[3 * GIC_MAX_LAYERS + 0] = B * L * H * a;
  out[3 * GIC_MAX_LAYERS + 1] = B * (uint64_t)c.V * 4;
  out[3 * GIC_MAX_LAYERS + 2] = B * 4 * H * 4;
  out[3 * GIC_MAX_LAYERS + 3] = (uint64_t)decoder_step_part_floats(c.B, c.L, c.V) * 4;
  return GIC_OK;
}

int gic_decoder_bwd_ws_bytes(const gic_decoder_dims* dims, uint64_t* out) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(out, "decoder_bwd_ws_bytes: null out");
  const uint64_t B = c.B, L = c.L, H = c.H, a = c.asz();
  out[0] = B * L * (uint64_t)c.V * a;
  out[1] = B * L * H * 4;
  for (int l = 0; l < GIC_MAX_LAYERS; ++l) {
    const bool on = l < c.NL;
    out[2 + l] = on ? L * B * 4 * H * a : 0;
    out[2 + GIC_MAX_LAYERS + l] = on ? (L + 1) * B * (uint64_t)c.ldx(l) * 4 : 0;
    out[2 + 2 * GIC_MAX_LAYERS + l] = on ? B * H * 4 : 0;
  }
  return GIC_OK;
}

int gic_decoder_prepare(const gic_decoder_dims* dims, const gic_decoder_params* P, const gic_decoder_shadow* S, void* stream_) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(P && S, "decoder_prepare: null struct");
  hipStream_t stream = (hipStream_t)stream_;
  for (int l = 0; l < c.NL; ++l) {
    GIC_CHECK_ARG(P->w_ih[l] && P->w_hh[l] && P->b_ih[l] && P->b_hh[l] && S->wcat[l] && S->bsum[l], "decoder_prepare: null layer %d", l);
    const long total = (long)4 * c.H * c.ldx(l);
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if (c.dt == DT_F32)
      hipLaunchKernelGGL((build_wcat_kernel<float>), dim3(grid), dim3(256), 0, stream, P->w_ih[l], P->w_hh[l], P->b_ih[l],
                         P->b_hh[l], (float*)S->wcat[l], S->bsum[l], 4 * c.H, c.din(l), c.H);
    else
      hipLaunchKernelGGL((build_wcat_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, P->w_ih[l], P->w_hh[l], P->b_ih[l],
                         P->b_hh[l], (bf16_t*)S->wcat[l], S->bsum[l], 4 * c.H, c.din(l), c.H);
    GIC_CHECK_LAUNCH("build_wcat");
    if (S->wcat_t[l]) GIC_PROPAGATE(transpose2d(S->wcat[l], S->wcat_t[l], c.dt, 4 * c.H, c.ldx(l), stream));
  }
  GIC_CHECK_ARG(P->w_out && S->wout, "decoder_prepare: null w_out");
  if ((const void*)S->wout != (const void*)P->w_out)
    GIC_PROPAGATE(cast2d(P->w_out, DT_F32, c.H, S->wout, c.dt, c.H, c.V, c.H, stream));
  return GIC_OK;
}

int gic_decoder_sample_fwd(const gic_decoder_dims* dims, const gic_decoder_params* P, const gic_decoder_shadow* S,
                           const gic_decoder_state* st, const float* features, const float* noise_u, uint64_t seed,
                           float temperature, int pretrain, void* out, int64_t* ids, const gic_decoder_sample_opts* opt,
                           void* stream) {
  Ctx c;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(P && S && st && features && ids, "decoder_sample_fwd: null argument");
  const bool keep = !(opt && opt->no_state);
  GIC_CHECK_ARG(!keep || (st->hout && out), "decoder_sample_fwd: out / state->hout may be NULL only for a stateless roll-out (opts->no_state)");
  GIC_CHECK_ARG(!(opt && opt->force_len && !opt->force_ids), "decoder_sample_fwd: force_len without force_ids");
  GIC_CHECK_ARG(!(opt && opt->dev_scalars) || (opt->seed_slot >= 0 && opt->seed_slot < GIC_STEP_SEEDS), "decoder_sample_fwd: seed_slot out of range");
  if (opt && opt->resume_from) {
    GIC_CHECK_ARG(opt->force_ids && opt->force_len && opt->host_active_rows && opt->resume_B > 0 && opt->no_state,
                  "decoder_sample_fwd: resumed roll-outs need force_ids, force_len, host_active_rows, resume_B and no_state");
    for (int l = 0; l < c.NL; ++l) GIC_CHECK_ARG(opt->resume_from->xh[l] && opt->resume_from->c[l], "decoder_sample_fwd: resume_from lacks layer %d state", l);
    for (int t = 0; t < c.L; ++t)
      GIC_CHECK_ARG(opt->host_active_rows[t] >= (t ? opt->host_active_rows[t - 1] : 0) && opt->host_active_rows[t] <= c.B,
                    "decoder_sample_fwd: host_active_rows must be non-decreasing and <= B");
    GIC_CHECK_ARG(opt->host_active_rows[0] == 0, "decoder_sample_fwd: resumed rows have a prefix of at least one token");
  }
  for (int l = 0; l < c.NL; ++l)
    GIC_CHECK_ARG(st->xh[l] && (st->gates[l] || !keep) && st->c[l] && S->wcat[l] && S->bsum[l], "decoder_sample_fwd: null layer %d buffer", l);
  if (c.dt == DT_F32)
    return sample_fwd_t<float>(c, P, S, st, features, noise_u, seed, temperature, pretrain, out, ids, opt, (hipStream_t)stream);
  return sample_fwd_t<bf16_t>(c, P, S, st, features, noise_u, seed, temperature, pretrain, out, ids, opt, (hipStream_t)stream);
}

int gic_decoder_sample_bwd(const gic_decoder_dims* dims, const gic_decoder_params* P, const gic_decoder_shadow* S,
                           const gic_decoder_state* st, const gic_decoder_bwd_ws* ws, const void* probs,
                           const int64_t* ids, const void* d_out, float temperature, int pretrain,
                           const gic_decoder_grads* G, int phases, const gic_step_scalars* dev_scalars, void* stream_) {
  Ctx c;
  const float* t_dev = dev_scalars ? &dev_scalars->temperature : nullptr;
  GIC_PROPAGATE(check_dims(dims, c));
  GIC_CHECK_ARG(P && S && st && ws && probs && ids && d_out && G, "decoder_sample_bwd: null argument");
  GIC_CHECK_ARG(ws->dlogits && ws->dhout && G->embed && G->w_out && G->b_out && G->features, "decoder_sample_bwd: null buffer");
  for (int l = 0; l < c.NL; ++l)
    GIC_CHECK_ARG(ws->dgates[l] && ws->dxh[l] && ws->dc[l] && G->w_ih[l] && G->w_hh[l] && G->b_ih[l] && G->b_hh[l],
                  "decoder_sample_bwd: null layer %d buffer", l);
  hipStream_t stream = (hipStream_t)stream_;
  GIC_CHECK_ARG(phases >= 1 && phases <= (GIC_DECODER_BWD_ALL | GIC_DECODER_BWD_STATE_GRADS), "decoder_sample_bwd: phases must be a mask of 1 | 2 | 4");
  int s = (c.dt == DT_F32)
              ? sample_bwd_t<float>(c, P, S, st, ws, probs, ids, d_out, temperature, t_dev, pretrain, G, phases, stream)
              : sample_bwd_t<bf16_t>(c, P, S, st, ws, probs, ids, d_out, temperature, t_dev, pretrain, G, phases, stream);
  GIC_PROPAGATE(s);
  if (phases & GIC_DECODER_BWD_RECURRENT)
    GIC_PROPAGATE(embed_scatter_time((const float*)ws->dxh[0], c.ldx(0), ids, G->embed, c.B, c.L, c.E, c.V, stream));
  return GIC_OK;
}

}  // extern "C"
