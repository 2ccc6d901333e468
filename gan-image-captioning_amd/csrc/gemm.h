// Host-side descriptor of the one MFMA GEMM family every dense contraction of
// the hot path goes through (kernels in gemm.hip).
//
//   C[m,n] = epi( alpha * sum_k A(m,k) * B(n,k) )
//
// A(m,k) = a_kc ? A[m*lda + k] : A[k*lda + m]      (k-contiguous / m-contiguous)
// B(n,k) = b_kc ? B[n*ldb + k] : B[k*ldb + n]
// so  forward  y = x W^T      : a_kc=1, b_kc=1   ("NT")
//     dgrad    dx = dy W      : a_kc=1, b_kc=0   ("NN")
//     wgrad    dW = dy^T x    : a_kc=0, b_kc=0   ("TN")
#pragma once
#include "common.h"

namespace gic {

enum GemmEpi { EPI_PLAIN = 0, EPI_HIGHWAY = 1, EPI_BNSTATS = 2, EPI_GUMBELMAX = 3 };

struct GemmDesc {
  const void* A = nullptr;
  const void* B = nullptr;
  void* C = nullptr;
  int M = 0, N = 0, K = 0;
  long lda = 0, ldb = 0, ldc = 0;
  int a_kc = 1, b_kc = 1;
  int in_dtype = DT_F32, out_dtype = DT_F32;
  const float* bias = nullptr;   // per output column n (optional)
  int accumulate = 0;            // C += result
  int c_zeroed = 0;              // caller guarantees C is all zero: a split-K launch skips its own zero fill
  float alpha = 1.f;
  int epi = EPI_PLAIN;
  // ---- EPI_HIGHWAY (discriminator.py:53-58): h = acc+bias; y = sig(h)*relu(h) + (1-sig(h))*x; C = y*keep*keep_scale
  const void* X = nullptr; long ldx = 0;        // carry input, in_dtype
  float* Hpre = nullptr; long ldh = 0;          // pre-activation h (saved for backward)
  const uint8_t* mask = nullptr; long ldmask = 0;   // explicit keep mask (0/1) or null
  uint8_t* mask_out = nullptr; long ldmask_out = 0;   // keep mask actually used, written for backward (optional)
  float keep_scale = 1.f;                       // 1/(1-p) in train mode, 1 in eval
  int use_philox = 0; float drop_p = 0.f; uint64_t seed = 0, stream = 0;
  const uint64_t* seed_dev = nullptr;           // non-null: the Philox seed is read from device memory (gic_step_scalars), `seed` is ignored
  // ---- implicit-GEMM convolution (conv != 0): A(m,k) is gathered from an NHWC activation `A`
  //      [Nimg, cH, cW, cCin] with m = (n, ho, wo) and k = (r, s, c); M = Nimg*cHo*cWo, K = cKH*cKW*cCin;
  //      B = weights [Cout, cKH, cKW, cCin] (k-contiguous).  Out-of-image taps read as zero.
  int conv = 0;
  int cH = 0, cW = 0, cCin = 0, cHo = 0, cWo = 0, cKH = 0, cKW = 0, cStride = 1, cPad = 0;
  // ---- EPI_BNSTATS: C = result (+bias) and stats[n] += sum_m v, stats[N+n] += sum_m v^2 (f32 atomics)
  float* stats = nullptr;   // [stats_nrep][2N]; block b adds into replica b % stats_nrep (readers sum the replicas)
  int stats_nrep = 1;
  // ---- A-side BatchNorm + ReLU (tile8 convolutions with Cin % 8 == 0, Cin <= 1024): the input activation is read as
  //      relu(scale[c] * x + shift[c]) (zero padding applied AFTER it, as the reference pads the normalised tensor) with
  //      scale / shift from the producer's batch statistics in_stats [in_nrep][2 Cin] (sum, sum of squares over
  //      in_inv_count^-1 rows), in_gamma, in_beta.  The producer's raw output is consumed directly: its separate
  //      normalisation pass (and the normalised tensor) disappear.
  const float* in_stats = nullptr; int in_nrep = 1;
  const float* in_gamma = nullptr; const float* in_beta = nullptr;
  float in_inv_count = 0.f;
  // ---- ... + residual (the block output of a ResNet bottleneck formed on load; 1x1 / stride 1 / pad 0 consumers only): the operand is
  //      relu(scale*x + shift + r) with r = res[m, c] as is (identity shortcut) or res_scale*res + res_shift from res_stats / res_gamma /
  //      res_beta (projection shortcut, its own BatchNorm).  Workgroups of the first N tile also WRITE the formed tile to out_wb
  //      [M, Cin] (the block output, needed again as the next shortcut): the separate bn + add + relu pass disappears.
  const void* res = nullptr;
  const float* res_stats = nullptr; int res_nrep = 1;
  const float* res_gamma = nullptr; const float* res_beta = nullptr;
  float res_inv_count = 0.f;
  void* out_wb = nullptr;
  // ---- EPI_GUMBELMAX (gemm_gumbelmax below): C is NOT written.  Rows m = vocabulary entries (A = W_out [V, H]), columns n = roll-out
  //      rows (B = their hidden states): key[n] = atomicMax over m of row_key((acc + gm_bias[m] + gumbel(u[n, m])) * gm_temperature, m)
  //      -- the token of an ids-only roll-out step (generator.py:68-73 with the softmax skipped: it is monotone) without the [rows, V]
  //      logits ever reaching memory.  u: gm_u (explicit uniforms, row n at gm_u + n * gm_ldu) or Philox(seed | seed_dev, stream) indexed
  //      as the unfused kernels index it (4 consecutive vocabulary entries of a row per call).
  unsigned long long* gm_rowkey = nullptr;
  const float* gm_bias = nullptr;
  const float* gm_u = nullptr; long gm_ldu = 0;
  float gm_temperature = 1.f;
  int n_fast = 0;           // tile8: the output-channel tiles of a row tile are neighbours on one XCD (xcd_share_a) instead of the row tiles of a channel tile
  int stats_only = 0;       // EPI_BNSTATS convolutions the streaming 1x1 kernel takes: the column sums only, C is NOT written (conv_b2b.hip's first pass);
                            // honoured by conv1x1_stream.hip alone: gemm() answers GIC_ERR_UNSUPPORTED when that kernel declines the shape
  int dbg = 0;              // phase-ablation knob, honoured only by -DGIC_STAMPS tool builds
};

// Enqueue on `stream`. Returns GIC_OK or a negative Status (message via gic_last_error()).
int gemm(const GemmDesc& d, hipStream_t stream);

// The vocabulary product of an ids-only roll-out step fused with Gumbel-max (EPI_GUMBELMAX above): bf16 k-contiguous operands, M = V a
// multiple of 4, many roll-out rows (the 8-wave kernel's grid conditions).  GIC_ERR_UNSUPPORTED (no message) when the shapes do not
// qualify: the caller runs the product and the Gumbel-argmax kernel separately.
int gemm_gumbelmax(const GemmDesc& d, hipStream_t stream);

}  // namespace gic
