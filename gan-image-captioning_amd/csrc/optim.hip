// GAN losses (reference src/utils.py:10-53), cross-entropy of the MLE pre-train step
// (src/training.py:81-83) and the fused clip_grad_norm_ + Adam update over a flat parameter
// arena (src/training.py:194-199).  All memory-bound; reductions use wavefront (64-lane)
// shuffles and one LDS hop per block, and are deterministic (fixed partial order).
#include "../../include/gicap.h"
#include "kernels.h"

namespace gic {
namespace {

__device__ __forceinline__ float softplusf_(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigm_(float z) { return 1.f / (1.f + expf(-z)); }

// One block: means over n logits + elementwise gradients (upstream gradient = 1 for each loss).
__global__ __launch_bounds__(1024) void gan_losses_kernel(int type, const float* __restrict__ dr, const float* __restrict__ df,
                                                           const float* __restrict__ go, long n, float* __restrict__ losses,
                                                           float* __restrict__ dd_real, float* __restrict__ dd_fake,
                                                           float* __restrict__ dg_out, float* __restrict__ dg_real,
                                                           float* __restrict__ dg_fake) {
  __shared__ float red[16];
  const float inv = 1.f / (float)n;
  float sd = 0.f, sg = 0.f;
  // 4 strided elements per pass with their 12 loads issued together (n = B*R = 4096: one pass of the 1024 threads)
  for (long i0 = threadIdx.x; i0 < n; i0 += 4L * blockDim.x) {
   float rr[4], ff[4], gg[4];
#pragma unroll
   for (int q = 0; q < 4; ++q) {
     const long i = i0 + (long)q * blockDim.x;
     rr[q] = i < n ? dr[i] : 0.f; ff[q] = i < n ? df[i] : 0.f; gg[q] = i < n ? go[i] : 0.f;
   }
#pragma unroll
   for (int q = 0; q < 4; ++q) {
    const long i = i0 + (long)q * blockDim.x;
    if (i >= n) break;
    const float r = rr[q], f = ff[q], g = gg[q];
    float ld = 0.f, lg = 0.f, gdr = 0.f, gdf = 0.f, ggo = 0.f, ggr = 0.f, ggf = 0.f;
    switch (type) {
      case GIC_LOSS_STANDARD:
      case GIC_LOSS_JS:
      case GIC_LOSS_KL:
        ld = softplusf_(-r) + softplusf_(f);           // BCE(d_real,1) + BCE(d_fake,0)
        gdr = sigm_(r) - 1.f; gdf = sigm_(f);
        if (type == GIC_LOSS_STANDARD) { lg = softplusf_(-g); ggo = sigm_(g) - 1.f; }
        else if (type == GIC_LOSS_JS) { lg = -softplusf_(g); ggo = -sigm_(g); }
        else { lg = -g; ggo = -1.f; }
        break;
      case GIC_LOSS_HINGE:
        ld = fmaxf(1.f - r, 0.f) + fmaxf(1.f + f, 0.f);
        gdr = (1.f - r) > 0.f ? -1.f : 0.f; gdf = (1.f + f) > 0.f ? 1.f : 0.f;
        lg = -g; ggo = -1.f;
        break;
      case GIC_LOSS_TV: {
        const float tr = tanhf(r), tf = tanhf(f), tg = tanhf(g);
        ld = tf - tr; gdr = -(1.f - tr * tr); gdf = 1.f - tf * tf;
        lg = -tg; ggo = -(1.f - tg * tg);
      } break;
      case GIC_LOSS_RSGAN:
        ld = softplusf_(-(r - f)); gdr = sigm_(r - f) - 1.f; gdf = -gdr;
        lg = softplusf_(-(f - r)); ggf = sigm_(f - r) - 1.f; ggr = -ggf;
        break;
      default: break;
    }
    sd += ld; sg += lg;
    if (dd_real) dd_real[i] = gdr * inv;
    if (dd_fake) dd_fake[i] = gdf * inv;
    if (dg_out) dg_out[i] = ggo * inv;
    if (dg_real) dg_real[i] = ggr * inv;
    if (dg_fake) dg_fake[i] = ggf * inv;
   }
  }
  sd = block_sum(sd, red);
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) { losses[0] = sg * inv; losses[1] = sd * inv; }
}

// CrossEntropyLoss (mean over all rows, no ignore_index): per-row block; loss partial per row, summed by a tail kernel
template <typename TA>
__global__ __launch_bounds__(256) void xent_rows_kernel(const TA* __restrict__ logits, long rows, int V,
                                                         const int64_t* __restrict__ targets, float* __restrict__ row_loss,
                                                         TA* __restrict__ dlogits, const float* __restrict__ row_weight) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const TA* x = logits + row * V;
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, to_f32<TA>(x[v]));
  mx = block_max(mx, red);
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) s += expf(to_f32<TA>(x[v]) - mx);
  s = block_sum(s, red);
  long tgt = targets[row];
  // nn.CrossEntropyLoss raises on a target outside [0, V) (training.py:81-83 has no ignore_index): no exception can cross a
  // kernel, so such a row poisons the mean with NaN (the host loop raises on it) instead of silently training on a clamped label
  const bool bad = tgt < 0 || tgt >= V;
  tgt = bad ? 0 : tgt;
  const float lse = mx + logf(s);
  const float wgt = row_weight ? row_weight[row] : 1.f;       // REINFORCE: the row's reward (policy-gradient loss); 1 for plain NLL
  if (threadIdx.x == 0) row_loss[row] = bad ? NAN : wgt * (lse - to_f32<TA>(x[tgt]));
  if (dlogits) {
    const float inv_rows = wgt / (float)rows;
    for (int v = threadIdx.x; v < V; v += 256) {
      const float p = expf(to_f32<TA>(x[v]) - mx) / s;
      dlogits[row * V + v] = from_f32<TA>((p - (v == tgt ? 1.f : 0.f)) * inv_rows);
    }
  }
}
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// ---- Monte-Carlo roll-out rewards (SeqGAN): reward[b, t] = mean over the N roll-outs of prefix length t+1 (and the R representation
// logits of each) of sigmoid(D logit); the last position is scored on the complete caption itself.  One block per (b, t).
__global__ __launch_bounds__(256) void rollout_rewards_kernel(const float* __restrict__ mc_logits, const float* __restrict__ full_logits,
                                                               float* __restrict__ rewards, int B, int L, int N, int R) {
  __shared__ float red[16];
  const int b = blockIdx.x / L, t = blockIdx.x % L;
  float s = 0.f;
  if (t + 1 < L) {
    for (int i = threadIdx.x; i < N * R; i += 256) {
      const int n = i / R, r = i % R;
      s += 1.f / (1.f + expf(-mc_logits[(((long)t * N + n) * B + b) * R + r]));
    }
    s = block_sum(s, red) / (float)(N * R);
  } else {
    for (int r = threadIdx.x; r < R; r += 256) s += 1.f / (1.f + expf(-full_logits[(long)b * R + r]));
    s = block_sum(s, red) / (float)R;
  }
  if (threadIdx.x == 0) rewards[(long)b * L + t] = s;
}

// ---- clip + Adam -----------------------------------------------------------------------------
constexpr int kNormBlock = 256;
constexpr long kNormElemsPerBlock = 256 * 16;

__global__ __launch_bounds__(kNormBlock) void sqnorm_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partials,
                                                                     int64_t* __restrict__ step_count) {
  __shared__ float red[16];
  const long base = (long)blockIdx.x * kNormElemsPerBlock;
  float s = 0.f;
  if (base + kNormElemsPerBlock <= n && (((uintptr_t)g) & 15) == 0) {
    // whole block: four 16-byte loads per thread, all in flight (the scalar loop below keeps one 4-byte load in flight)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* gv = (const f4*)(g + base);
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = gv[threadIdx.x + kNormBlock * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
  } else {
    for (long i = base + threadIdx.x; i < base + kNormElemsPerBlock && i < n; i += kNormBlock) { const float v = g[i]; s += v * v; }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = s;
    if (blockIdx.x == 0) step_count[0] += 1;       // optimizer step counter lives on the device (graph-replay safe)
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long n, double lr_d, double b1_d, double b2_d, double eps_d,
                                                         double clip_d, const int64_t* __restrict__ step_count,
                                                         const float* __restrict__ partials, int nparts, float* __restrict__ norm_out) {
  __shared__ float red[16];
  // every block re-reduces the (few) partials in the same order -> identical clip coefficient everywhere
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
  s = block_sum(s, red);
  const float norm = sqrtf(s);
  if (blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
  float coef = (float)clip_d / (norm + 1e-6f);       // torch.nn.utils.clip_grad_norm_
  coef = coef > 1.f ? 1.f : coef;
  // scalar hyper-parameter arithmetic in float64, like torch.optim.Adam's Python scalars
  const double t = (double)step_count[0];
  const float omb1 = (float)(1.0 - b1_d), omb2 = (float)(1.0 - b2_d), b2 = (float)b2_d, eps = (float)eps_d;
  const float bc2_sqrt = (float)sqrt(1.0 - pow(b2_d, t));
  const float step_size = (float)(lr_d / (1.0 - pow(b1_d, t)));
  auto update = [&](float gi, float& mi, float& vi, float& pi) {
    gi *= coef;
    mi = mi + (gi - mi) * omb1;                              // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * b2 + omb2 * gi * gi;                           // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
  };
  // 16-byte accesses over the aligned body (the four arenas share their alignment: same offsets), scalars over the tail
  typedef float f4 __attribute__((ext_vector_type(4)));
  const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  const long n4 = al ? n / 4 : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f4 g4 = ((const f4*)g)[i];
    f4 m4 = ((f4*)m)[i], v4 = ((f4*)v)[i], p4 = ((f4*)p)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) { float mi = m4[k], vi = v4[k], pi = p4[k]; update(g4[k], mi, vi, pi); m4[k] = mi; v4[k] = vi; p4[k] = pi; }
    ((f4*)m)[i] = m4; ((f4*)v)[i] = v4; ((f4*)p)[i] = p4;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float mi = m[i], vi = v[i], pi = p[i];
    update(g[i], mi, vi, pi);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

}  // namespace
}  // namespace gic

using namespace gic;

extern "C" {

int gic_gan_losses(int loss_type, const float* d_real, const float* d_fake, const float* g_out, int64_t n, float* losses,
                   float* dd_real, float* dd_fake, float* dg_out, float* dg_real, float* dg_fake, void* stream) {
  GIC_CHECK_ARG(loss_type >= GIC_LOSS_STANDARD && loss_type <= GIC_LOSS_RSGAN, "gan_losses: unknown loss type %d", loss_type);
  GIC_CHECK_ARG(d_real && d_fake && g_out && losses && n > 0, "gan_losses: null pointer or n<=0");
  hipLaunchKernelGGL(gan_losses_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_type, d_real, d_fake, g_out, (long)n, losses,
                     dd_real, dd_fake, dg_out, dg_real, dg_fake);
  GIC_CHECK_LAUNCH("gan_losses");
  return GIC_OK;
}

// `loss`: device f32[1 + rows]: loss[0] = mean, loss[1..] = per-row scratch.
int gic_xent(const void* logits, int dtype, int64_t rows, int32_t V, const int64_t* targets, float* loss, void* d_logits,
             const float* row_weight, void* stream_) {
  GIC_CHECK_ARG(logits && targets && loss && rows > 0 && V > 0, "xent: bad argument");
  hipStream_t stream = (hipStream_t)stream_;
  if (dtype == DT_F32)
    hipLaunchKernelGGL((xent_rows_kernel<float>), dim3((unsigned)rows), dim3(256), 0, stream, (const float*)logits, (long)rows, V, targets,
                       loss + 1, (float*)d_logits, row_weight);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL((xent_rows_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)logits, (long)rows, V, targets,
                       loss + 1, (bf16_t*)d_logits, row_weight);
  else { set_last_error("xent: bad dtype %d", dtype); return GIC_ERR_UNSUPPORTED; }
  GIC_CHECK_LAUNCH("xent_rows");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, stream, (const float*)(loss + 1), (long)rows, loss);
  GIC_CHECK_LAUNCH("xent_mean");
  return GIC_OK;
}

int gic_rollout_rewards(const float* mc_logits, const float* full_logits, float* rewards, int B, int L, int N, int R, void* stream) {
  GIC_CHECK_ARG(full_logits && rewards && B > 0 && L > 0 && R > 0 && (L == 1 || (mc_logits && N > 0)), "rollout_rewards: bad argument");
  hipLaunchKernelGGL(rollout_rewards_kernel, dim3((unsigned)(B * L)), dim3(256), 0, (hipStream_t)stream, mc_logits, full_logits, rewards, B, L, N, R);
  GIC_CHECK_LAUNCH("rollout_rewards");
  return GIC_OK;
}

int64_t gic_clip_adam_partials(int64_t n) { return (n + kNormElemsPerBlock - 1) / kNormElemsPerBlock; }

int gic_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, double clip_norm, int64_t* step_count, float* norm_out, float* partials, void* stream_) {
  GIC_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && step_count && norm_out && partials, "clip_adam: null pointer");
  GIC_CHECK_ARG(n > 0, "clip_adam: n<=0");
  hipStream_t stream = (hipStream_t)stream_;
  const int nparts = (int)gic_clip_adam_partials(n);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nparts), dim3(kNormBlock), 0, stream, grads, (long)n, partials, step_count);
  GIC_CHECK_LAUNCH("sqnorm_partial");
  long blocks = (n + 256 * 8 - 1) / (256 * 8);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, (long)n, lr,
                     beta1, beta2, eps, clip_norm, (const int64_t*)step_count, (const float*)partials, nparts, norm_out);
  GIC_CHECK_LAUNCH("clip_adam");
  return GIC_OK;
}

}  // extern "C"
