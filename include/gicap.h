/* gicap.h - C ABI of libgicap.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * adversarial image-captioning train step.
 *
 * The reference (kawshik8/GAN-Image-Captioning) has no FFI: its boundary for this
 * path is the Python module API consumed by src/training.py.  Each entry point below
 * therefore cites the reference interface whose compute it replaces (file:line under
 * /root/reference/); the Python host in gan-image-captioning_amd/ keeps the module
 * API and binds these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory unless named host_*.
 *  - the caller owns every buffer (parameters, outputs, saved-for-backward state,
 *    workspaces); the library allocates nothing and keeps no global state except
 *    the thread-local last-error string.
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and returns
 *    immediately: 0 = OK, negative = gic_status.  No exceptions, no exit().
 *  - dtype: GIC_F32 (parity mode; exact-f32 MFMA) or GIC_BF16 (bf16 MFMA operands,
 *    f32 accumulation, f32 master weights / grads / optimizer state).
 *    "act" buffers below have the compute dtype; everything else is f32.
 */
#ifndef GICAP_H_
#define GICAP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIC_ABI_VERSION 4
#define GIC_MAX_LAYERS 4
#define GIC_MAX_CONVS 8

enum gic_status {
  GIC_STATUS_OK = 0,
  GIC_STATUS_INVALID_ARG = -1,
  GIC_STATUS_UNSUPPORTED = -2,
  GIC_STATUS_LAUNCH = -3,
  GIC_STATUS_WORKSPACE = -4
};
enum gic_dtype { GIC_F32 = 0, GIC_BF16 = 1 };
enum gic_loss_type {           /* src/utils.py:10-53 */
  GIC_LOSS_STANDARD = 0, GIC_LOSS_JS = 1, GIC_LOSS_KL = 2, GIC_LOSS_HINGE = 3, GIC_LOSS_TV = 4, GIC_LOSS_RSGAN = 5
};

int gic_abi_version(void);
/* Message of the last failing call on this host thread ("" if none). */
const char* gic_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Per-step scalars resident in DEVICE memory (ABI v3).  The values that change from one train step to the next -- the decoder's
 * temperature (src/training.py:183, 190-191: updated after every batch) and the seeds of the device noise streams (Gumbel uniforms,
 * src/generator.py:86-90; the three dropout draws, src/discriminator.py:58) -- can be read by the kernels from this struct instead
 * of being passed by value: the launch arguments of a whole step then never change, which is what lets the step be ONE replayed
 * hipGraph.  Entry points that take a `dev_scalars` pointer ignore their by-value `temperature` / `seed` when it is non-NULL and
 * read dev_scalars->temperature / dev_scalars->seed[seed_slot] on the device.  gic_step_scalars_set enqueues a one-thread kernel
 * that writes *dev from the host values (captured by value at the call: no pinned staging buffer to keep alive). */
#define GIC_STEP_SEEDS 6
typedef struct gic_step_scalars {
  float temperature;
  uint32_t reserved;
  uint64_t seed[GIC_STEP_SEEDS];
} gic_step_scalars;
int gic_step_scalars_set(gic_step_scalars* dev, const gic_step_scalars* host_values, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generic dense contraction (used by the tests and by every entry point below).
 *   C[m,n] = alpha * sum_k A(m,k) B(n,k) + bias[n]  (+ C if accumulate)
 *   A(m,k) = a_kc ? A[m*lda+k] : A[k*lda+m];  B(n,k) = b_kc ? B[n*ldb+k] : B[k*ldb+n]
 * Replaces the ATen GEMMs behind nn.Linear / nn.LSTM / autograd (generator.py:61,68;
 * discriminator.py:40,53,58,60).
 */
int gic_gemm(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
             int a_kc, int b_kc, int in_dtype, int out_dtype, const float* bias, int accumulate, float alpha,
             void* stream);

/* 2-D copy with dtype conversion: dst[r*ldd + c] = src[r*lds + c]. */
int gic_cast2d(const void* src, int src_dtype, int64_t lds, void* dst, int dst_dtype, int64_t ldd,
               int64_t rows, int64_t cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generator: Decoder.sample (src/generator.py:55-96) forward + backward.
 */
typedef struct gic_decoder_dims {
  int32_t B, L, V, E, H, NL;   /* batch, max_caption_len, vocab, gen_embed_dim, gen_hidden_dim, gen_num_layers */
  int32_t dtype;               /* compute dtype */
} gic_decoder_dims;

typedef struct gic_decoder_params {      /* f32 master weights, state-dict layout of nn.Embedding/nn.LSTM/nn.Linear */
  const float* embed;                    /* decoder.embed.weight            [V,E]      */
  const float* w_ih[GIC_MAX_LAYERS];     /* decoder.lstm.weight_ih_l{k}     [4H,Din_k] */
  const float* w_hh[GIC_MAX_LAYERS];     /* decoder.lstm.weight_hh_l{k}     [4H,H]     */
  const float* b_ih[GIC_MAX_LAYERS];     /* decoder.lstm.bias_ih_l{k}       [4H]       */
  const float* b_hh[GIC_MAX_LAYERS];     /* decoder.lstm.bias_hh_l{k}       [4H]       */
  const float* w_out;                    /* decoder.linear.weight           [V,H]      */
  const float* b_out;                    /* decoder.linear.bias             [V]        */
} gic_decoder_params;

typedef struct gic_decoder_grads {       /* f32, same shapes as gic_decoder_params; all overwritten */
  float* embed;
  float* w_ih[GIC_MAX_LAYERS];
  float* w_hh[GIC_MAX_LAYERS];
  float* b_ih[GIC_MAX_LAYERS];
  float* b_hh[GIC_MAX_LAYERS];
  float* w_out;
  float* b_out;
  float* features;                       /* d(features) [B,E] */
} gic_decoder_grads;

/* Derived per-step weight images in the compute dtype (caller-owned, refreshed by gic_decoder_prepare). */
typedef struct gic_decoder_shadow {
  void* wcat[GIC_MAX_LAYERS];            /* act [4H, Din_k+H] = [w_ih | w_hh] */
  float* bsum[GIC_MAX_LAYERS];           /* [4H] = b_ih + b_hh */
  void* wout;                            /* act [V,H] (may alias params.w_out in f32 mode) */
  void* wcat_t[GIC_MAX_LAYERS];          /* act [Din_k+H, 4H] = Wcat^T, the k-contiguous operand of the BPTT input-gradient
                                            product (the fused BPTT step kernel needs it); NULL: that product reads Wcat
                                            transposed instead, one generic product + one pointwise launch per step */
} gic_decoder_shadow;

/* Saved-for-backward state + scratch of one sample() call (caller-owned). Din_0=E, Din_k=H. */
typedef struct gic_decoder_state {
  void* xh[GIC_MAX_LAYERS];              /* act [(L+1), B, Din_k+H]: LSTM input | previous hidden, time-major */
  float* gates[GIC_MAX_LAYERS];          /* [L, B, 4H] post-activation i,f,g,o */
  float* c[GIC_MAX_LAYERS];              /* [(L+1), B, H] cell state, slot 0 = zeros */
  void* hout;                            /* act [B, L, H] last layer's h, batch-major (vocab GEMM operand) */
  float* logits;                         /* scratch [B, V] */
  float* gpre;                           /* scratch [B, 4H] */
  float* part;                           /* scratch of the fused step kernels: [2][L][B][ceil(V/64)] per-tile softmax partials (max, sum of
                                            exp) + [L][B] 64-bit argmax keys + two reserved 32-bit words; size from gic_decoder_state_bytes,
                                            8-byte aligned.
                                            NULL selects the unfused launches. */
} gic_decoder_state;

typedef struct gic_decoder_bwd_ws {      /* scratch for backward (caller-owned) */
  void* dlogits;                         /* act [B, L, V] */
  float* dhout;                          /* [B, L, H] */
  void* dgates[GIC_MAX_LAYERS];          /* act [L, B, 4H] */
  float* dxh[GIC_MAX_LAYERS];            /* [(L+1), B, Din_k+H] */
  float* dc[GIC_MAX_LAYERS];             /* [B, H] */
} gic_decoder_bwd_ws;

/* Byte size of every buffer of the caller-owned structs above for `dims`, in field order (per-layer arrays take GIC_MAX_LAYERS
 * entries, 0 for unused layers): gic_decoder_state -> xh[], gates[], c[], hout, logits, gpre, part (3*GIC_MAX_LAYERS + 4 values);
 * gic_decoder_bwd_ws -> dlogits, dhout, dgates[], dxh[], dc[] (2 + 3*GIC_MAX_LAYERS values).  Host-only: no GPU needed. */
int gic_decoder_state_bytes(const gic_decoder_dims* dims, uint64_t* out);
int gic_decoder_bwd_ws_bytes(const gic_decoder_dims* dims, uint64_t* out);

int gic_decoder_prepare(const gic_decoder_dims* dims, const gic_decoder_params* params,
                        const gic_decoder_shadow* shadow, void* stream);

/* Optional arguments of gic_decoder_sample_fwd (NULL = none). */
typedef struct gic_decoder_sample_opts {
  const float* h0;                       /* [NL,B,H] initial hidden state: sample(features, states=(h0, c0)), generator.py:55,61; NULL = zeros */
  const float* c0;                       /* [NL,B,H] initial cell state; NULL = zeros */
  const int64_t* force_ids;              /* [B,L] trajectory to FOLLOW: where forced, step t feeds embed(force_ids[b,t]) to step t+1 and
                                            returns it in ids (out is computed as usual).  Prefixes of Monte-Carlo roll-outs; parity tests. */
  const int32_t* force_len;              /* [B] number of leading steps that are forced per caption; NULL = all L */
  int32_t no_state;                      /* != 0: inference roll-out -- nothing is saved for a backward pass: state->gates / hout and
                                            `out` may be NULL (ids only) */
  /* Resumed roll-outs (Monte-Carlo completions of prefixes of ONE sampled batch; generic-product path, i.e. more rows than the fused
   * step kernels take): a row does not recompute its forced prefix.  Row r starts at step force_len[r] (>= 1) from the recurrent state
   * that the call which filled `resume_from` (same weights; caption r % resume_B; the same forced tokens) had reached there.  Rows must
   * be sorted by force_len ascending; host_active_rows[t] (HOST memory, L values) = number of rows with force_len <= t. */
  const struct gic_decoder_state* resume_from;
  int32_t resume_B;
  const int32_t* host_active_rows;
  /* device-resident temperature and Philox seed (gic_step_scalars above; fused step kernels only: GIC_STATUS_UNSUPPORTED on the
   * generic-product path) */
  const gic_step_scalars* dev_scalars;
  int32_t seed_slot;
} gic_decoder_sample_opts;

/* features [B,E] f32.  noise_u: explicit U[0,1) draws [L,B,V] f32 (generator.py:86-90 order) or NULL to
 * draw on device with Philox(seed, step).  pretrain != 0: generator.py:63-66 (out = raw logits, feedback =
 * argmax).  out: act [B,L,V] (probabilities or logits).  ids: int64 [B,L].
 * With state->part set and V % 4 == 0, E % 8 == 0, H % 8 == 0 a step is two fused launches (gates product + LSTM cell;
 * vocabulary product + Gumbel + per-tile softmax partials) and the probabilities are normalised by one launch at the end. */
int gic_decoder_sample_fwd(const gic_decoder_dims* dims, const gic_decoder_params* params,
                           const gic_decoder_shadow* shadow, const gic_decoder_state* state,
                           const float* features, const float* noise_u, uint64_t seed, float temperature,
                           int pretrain, void* out, int64_t* ids, const gic_decoder_sample_opts* opts, void* stream);

/* Tools only (tools/rollout_bench.py): phase-ablation mask of the fused step kernels; 0 = normal operation. */
void gic_debug_decoder_step(int mask);

/* Which roll-out path gic_decoder_sample_fwd takes for `dims` (host-only, no GPU needed): *out_rows = the largest batch B the fused
 * step kernels take for these V / E / H / NL (GIC_FUSED_ROLLOUT_MAX_ROWS, default 512), or 0 when they decline the shapes
 * (V % 4, E % 8, H % 8, GIC_NO_FUSED_ROLLOUT).  A call with B beyond that, with state->part == NULL or with opts->resume_from runs
 * the generic products and NEEDS state->logits and state->gpre; a caller that sizes its buffers by this query never trips that check
 * (Decoder.sample, src/generator.py:55-81, takes any vocabulary / embedding size). */
int gic_decoder_fused_rollout_rows(const gic_decoder_dims* dims, int32_t* out_rows);

/* Decoder.forward, the teacher-forced decode (src/generator.py:39-53; the reference's training never calls it).
 * dims->L = T = caption length + 1 time steps: step 0 is fed `features`, step t > 0 embed(caps[b, t-1]) (caps int64 [B, T-1]).
 * lengths int32 [B] (each 1..T) with pack_padded_sequence semantics: a row past its length keeps its state and contributes a
 * zero LSTM output.  Tmax = max(lengths) (the host knows it).  out: act [B, Tmax, V] = logits (pretrain != 0) or
 * softmax((logits + gumbel(u)) * temperature) with u = noise_u f32 [B, Tmax, V] (ONE draw over the whole tensor,
 * generator.py:50,86-90) or Philox(seed) when NULL.  h_n / c_n: f32 [NL, B, H], each row's state at ITS last step.
 * state: as for gic_decoder_sample_fwd with L = T; logits_ws f32 [B*Tmax, V] and ids_ws int64 [B*Tmax]: scratch.  With
 * state->gates[k] non-NULL the gate activations of every step are saved and gic_decoder_forward_tf_bwd can follow. */
int gic_decoder_forward_tf(const gic_decoder_dims* dims, const gic_decoder_params* params, const gic_decoder_shadow* shadow,
                           const gic_decoder_state* state, const float* features, const int64_t* caps, const int32_t* lengths,
                           int Tmax, const float* noise_u, uint64_t seed, float temperature, int pretrain, float* logits_ws,
                           int64_t* ids_ws, void* out, float* h_n, float* c_n, void* stream);

/* (ABI v4) autograd through Decoder.forward (src/generator.py:39-53): gradients of a loss on `out` of the gic_decoder_forward_tf
 * call that filled `state` (same dims, caps, lengths, Tmax, temperature, pretrain; state->gates given).  pred = that call's `out`,
 * d_pred: act [B, Tmax, V].  Padded positions (t >= lengths[b]) are pad_packed_sequence's zeros (generator.py:45): their
 * gradient reaches b_out only.  ws / grads as for gic_decoder_sample_bwd with L = Tmax (grads->embed: scatter-add over caps;
 * grads->features = d features).  The returned hidden (h_n, c_n) is not differentiated. */
int gic_decoder_forward_tf_bwd(const gic_decoder_dims* dims, const gic_decoder_params* params, const gic_decoder_shadow* shadow,
                               const gic_decoder_state* state, const gic_decoder_bwd_ws* ws, const void* pred, const int64_t* caps,
                               const int32_t* lengths, int Tmax, const void* d_pred, float temperature, int pretrain,
                               const gic_decoder_grads* grads, void* stream);

/* d_out: act [B,L,V] gradient w.r.t. `out`; probs = the forward's `out`.
 * phases (bit mask; GIC_DECODER_BWD_ALL = both, in this order):
 *   GIC_DECODER_BWD_OUTPUT     softmax/Gumbel backward, d_hout, and the COMPLETE gradients of the vocabulary projection
 *                              (grads->w_out, grads->b_out): a data-parallel caller can start all-reducing them here
 *   GIC_DECODER_BWD_RECURRENT  BPTT, LSTM weight gradients, d_features, embedding gradient */
#define GIC_DECODER_BWD_OUTPUT 1
#define GIC_DECODER_BWD_RECURRENT 2
#define GIC_DECODER_BWD_ALL 3
/* with GIC_DECODER_BWD_RECURRENT: also the gradient of the initial states (sample(states=...)): d h0 of layer k is left in the h
 * columns of ws->dxh[k] slot 0, d c0 in ws->dc[k] */
#define GIC_DECODER_BWD_STATE_GRADS 4
int gic_decoder_sample_bwd(const gic_decoder_dims* dims, const gic_decoder_params* params,
                           const gic_decoder_shadow* shadow, const gic_decoder_state* state,
                           const gic_decoder_bwd_ws* ws, const void* probs, const int64_t* ids,
                           const void* d_out, float temperature, int pretrain,
                           const gic_decoder_grads* grads, int phases, const gic_step_scalars* dev_scalars, void* stream);

/* nn.Embedding used as a callable (training.py:68,147): out[i,:] = weight[ids[i],:] and its scatter-add. */
int gic_embedding_fwd(const float* weight, const int64_t* ids, float* out, int64_t n, int32_t V, int32_t E, void* stream);
int gic_embedding_bwd(const float* d_out, const int64_t* ids, float* d_weight, int64_t n, int32_t V, int32_t E,
                      int zero_first, void* stream);

/* ------------------------------------------------------------------------------------------
 * Visual-attention caption decoder (BASELINE config 4; NO reference counterpart: the reference's decoder, src/generator.py:27-96,
 * sees the image only through the pooled feature).  The reference's roll-out loop (generator.py:55-81) with a Show-Attend-Tell soft
 * attention over the trunk's feature map in front of a one-layer LSTM; definition + CPU oracle: oracle/cpu_attention.py.
 *   fp_i = W_f a_i + b_f;  e_ti = w_a . tanh(fp_i + W_h h_{t-1});  alpha_t = softmax_i e_t;  z_t = sum_i alpha_ti a_i;
 *   LSTM input [x_t ; z_t];  logits, Gumbel, softmax, argmax feedback as gic_decoder_sample_fwd.
 */
typedef struct gic_attn_dims {
  int32_t B, L, V, E, H;       /* as gic_decoder_dims */
  int32_t C, P, A;             /* feature channels, positions (h*w) of the feature map, attention width */
  int32_t dtype;
} gic_attn_dims;

typedef struct gic_attn_params {         /* f32 master weights */
  const float* embed;                    /* decoder.embed.weight        [V,E]     */
  const float* w_ih;                     /* decoder.lstm.weight_ih_l0   [4H,E+C]  */
  const float* w_hh;                     /* decoder.lstm.weight_hh_l0   [4H,H]    */
  const float* b_ih; const float* b_hh;  /* [4H] */
  const float* w_out; const float* b_out;/* decoder.linear              [V,H],[V] */
  const float* w_f; const float* b_f;    /* decoder.attn.w_f / b_f      [A,C],[A] */
  const float* w_h;                      /* decoder.attn.w_h            [A,H]     */
  const float* w_a;                      /* decoder.attn.w_a            [A]       */
} gic_attn_params;

typedef struct gic_attn_grads {          /* f32, shapes of gic_attn_params; all overwritten */
  float* embed; float* w_ih; float* w_hh; float* b_ih; float* b_hh; float* w_out; float* b_out;
  float* w_f; float* b_f; float* w_h; float* w_a;
  float* features;                       /* d(features) [B,E] */
} gic_attn_grads;

typedef struct gic_attn_shadow {         /* compute-dtype weight images, refreshed by gic_attn_prepare */
  void* wcat;                            /* act [4H, E+C+H] = [w_ih | w_hh] */
  float* bsum;                           /* [4H] */
  void* wout;                            /* act [V,H] (may alias params.w_out in f32 mode) */
  void* wcat_t;                          /* act [E+C+H, 4H] */
  void* wf;                              /* act [A,C] */
  void* wh;                              /* act [A,H] */
} gic_attn_shadow;

typedef struct gic_attn_state {          /* saved for backward + scratch (caller-owned) */
  void* xh;                              /* act [(L+1), B, E+C+H]: x_t | z_t | h_{t-1} */
  float* gates;                          /* [L, B, 4H] */
  float* c;                              /* [(L+1), B, H] */
  void* hout;                            /* act [B, L, H] */
  float* part;                           /* scratch of the fused step kernels, as gic_decoder_state.part */
  void* fproj;                           /* act [B, P, A] */
  float* alpha;                          /* [L, B, P] attention weights */
  float* hproj;                          /* [L, B, A] W_h h_{t-1} */
} gic_attn_state;

typedef struct gic_attn_bwd_ws {
  void* dlogits;                         /* act [B, L, V] */
  float* dhout;                          /* [B, L, H] */
  void* dgates;                          /* act [L, B, 4H] */
  float* dc;                             /* [B, H] */
  float* dz;                             /* [B, C] */
  float* dalpha;                         /* [B, P] */
  float* dh_extra;                       /* [B, H] */
  void* dhproj;                          /* act [L, B, A] */
  float* dfproj;                         /* [B, P, A] */
  void* dfproj_act;                      /* act [B, P, A] (bf16 mode; may be NULL in f32 mode) */
  float* dwa_rows;                       /* [B, A] */
  float* dx;                             /* [L*B, E] */
} gic_attn_bwd_ws;

int gic_attn_prepare(const gic_attn_dims* dims, const gic_attn_params* params, const gic_attn_shadow* shadow, void* stream);
/* features f32 [B,E] (the encoder head's output, x_0); fmap act [B,P,C] (the trunk's last feature map, NHWC flattened; no gradient
 * flows into it: the trunk is frozen, generator.py:21).  noise_u / seed / temperature / pretrain / out / ids as
 * gic_decoder_sample_fwd.  V % 4 == 0 and E, H, C, A % 8 == 0.  h0 / c0: f32 [B, H] initial hidden / cell state (the `states`
 * argument of Decoder.sample's signature, generator.py:55,61) or NULL = zeros; they are constants of the backward pass (no
 * gradient is returned for them).  dev_scalars / seed_slot: temperature and seed from device memory (gic_step_scalars). */
int gic_attn_sample_fwd(const gic_attn_dims* dims, const gic_attn_params* params, const gic_attn_shadow* shadow,
                        const gic_attn_state* state, const float* features, const void* fmap, const float* noise_u, uint64_t seed,
                        float temperature, int pretrain, void* out, int64_t* ids, const float* h0, const float* c0,
                        const gic_step_scalars* dev_scalars, int seed_slot, void* stream);
int gic_attn_sample_bwd(const gic_attn_dims* dims, const gic_attn_params* params, const gic_attn_shadow* shadow,
                        const gic_attn_state* state, const gic_attn_bwd_ws* ws, const void* fmap, const void* probs,
                        const int64_t* ids, const void* d_out, float temperature, int pretrain, const gic_attn_grads* grads,
                        const gic_step_scalars* dev_scalars, void* stream);

/* ------------------------------------------------------------------------------------------
 * Discriminator.forward (src/discriminator.py:34-62) forward + backward.
 */
typedef struct gic_disc_dims {
  int32_t B, L, V, De, R, nconv;         /* De=disc_embed_dim, R=disc_num_rep, s = De/R */
  int32_t fsize[GIC_MAX_CONVS];          /* disc_filter_sizes */
  int32_t nfilt[GIC_MAX_CONVS];          /* disc_num_filters */
  int32_t F;                             /* sum(nfilt) */
  int32_t Fp;                            /* leading dim of the [B*R, F] activations (>= F, multiple of 8) */
  int32_t dtype;
  float drop_p;                          /* nn.Dropout p before feature2out (discriminator.py:10,30; default 0.2), 0 <= p < 1 */
} gic_disc_dims;

typedef struct gic_disc_params {
  const float* emb;                      /* embeddings.weight  [De,V] */
  const float* conv_w[GIC_MAX_CONVS];    /* convs.k.weight     [n_k,1,f_k,s] */
  const float* conv_b[GIC_MAX_CONVS];    /* convs.k.bias       [n_k] */
  const float* hw_w; const float* hw_b;  /* highway            [F,F],[F] */
  const float* f2o_w; const float* f2o_b;/* feature2out        [100,F],[100] */
  const float* o2l_w; const float* o2l_b;/* out2logits         [1,100],[1] */
} gic_disc_params;

typedef struct gic_disc_grads {          /* f32; accumulate != 0 adds to the existing contents */
  float* emb; float* conv_w[GIC_MAX_CONVS]; float* conv_b[GIC_MAX_CONVS];
  float* hw_w; float* hw_b; float* f2o_w; float* f2o_b; float* o2l_w; float* o2l_b;
} gic_disc_grads;

typedef struct gic_disc_shadow {         /* compute-dtype weight images (may alias params in f32 mode) */
  void* emb;                             /* act [De,V] */
  void* hw_w;                            /* act [F,F]  */
  void* f2o_w;                           /* act [100,F] */
  void* hw_w_t;                          /* act [Fp,Fp] = highway^T, the k-contiguous operand of the highway input-gradient
                                            product; NULL: that product reads hw_w transposed instead */
} gic_disc_shadow;

typedef struct gic_disc_state {          /* saved-for-backward of one forward call (caller-owned) */
  float* emb;                            /* [B*L, De] */
  void* pooled;                          /* act [B*R, Fp] conv+relu+max-over-time, row = b*R+r */
  uint8_t* argmax;                       /* [B*R, Fp] time index of the max */
  float* hpre;                           /* [B*R, Fp] highway pre-activation */
  uint8_t* keep;                         /* [B*R, Fp] dropout keep mask actually used (train mode) */
  void* ydrop;                           /* act [B*R, Fp] dropped highway output */
  float* feat;                           /* [B*R, 100] */
} gic_disc_state;

typedef struct gic_disc_bwd_ws {
  void* dfeat;                           /* act [B*R, 104] */
  void* dh;                              /* act [B*R, Fp] */
  float* dydrop;                         /* [B*R, Fp] */
  float* dpooled;                        /* [B*R, Fp] */
  void* demb;                            /* act [B*L, De] */
} gic_disc_bwd_ws;

/* Byte sizes in field order: gic_disc_state -> emb, pooled, argmax, hpre, keep, ydrop, feat (7 values; ydrop must start zeroed:
 * its pad columns are read); gic_disc_bwd_ws -> dfeat, dh, dydrop, dpooled, demb (5 values).  Host-only. */
int gic_disc_state_bytes(const gic_disc_dims* dims, uint64_t* out);
int gic_disc_bwd_ws_bytes(const gic_disc_dims* dims, uint64_t* out);

int gic_disc_prepare(const gic_disc_dims* dims, const gic_disc_params* params, const gic_disc_shadow* shadow, void* stream);

/* Exactly one of inp_soft (act [B*L, V], row stride ld_inp, rows in (b,l) order) and inp_ids (int64 [B,L];
 * the one-hot of training.py:158 evaluated as a gather) is non-NULL.
 * train != 0: dropout(dims->drop_p) with keep_mask (uint8 0/1 [B*R,F], row stride F) or, if NULL, Philox(seed) -- the seed read
 * from dev_scalars->seed[seed_slot] on the device when dev_scalars != NULL (gic_step_scalars).
 * train == 0 with state->argmax == NULL and / or state->hpre == NULL: a forward that no backward follows (reward evaluation of
 * the SeqGAN-style step): those buffers are not written.  logits: f32 [B*R]. */
int gic_disc_fwd(const gic_disc_dims* dims, const gic_disc_params* params, const gic_disc_shadow* shadow,
                 const gic_disc_state* state, const void* inp_soft, int64_t ld_inp, const int64_t* inp_ids,
                 int train, const uint8_t* keep_mask, uint64_t seed, float* logits, const gic_step_scalars* dev_scalars,
                 int seed_slot, void* stream);

/* A second forward on the SAME input as the pass that filled `src` (training.py:163-164 run D twice on gen_captions: only
 * the dropout draw differs): reuses src's pooled features and highway pre-activation, applies a fresh dropout mask
 * (keep_mask or Philox(seed)) and the feature2out / out2logits head.  Writes dst->keep, dst->ydrop, dst->feat and logits; for
 * the backward pass of this forward, dst->emb / pooled / argmax / hpre must alias src's buffers. */
int gic_disc_fwd_redrop(const gic_disc_dims* dims, const gic_disc_params* params, const gic_disc_shadow* shadow,
                        const gic_disc_state* src, const gic_disc_state* dst, int train, const uint8_t* keep_mask,
                        uint64_t seed, float* logits, const gic_step_scalars* dev_scalars, int seed_slot, void* stream);

/* d_logits f32 [B*R].  grads may be NULL (no parameter gradients wanted: the generator's path,
 * training.py:169).  d_inp: act [B*L, V] (row stride ld_dinp) or NULL.
 * Exactly one of inp_soft / inp_ids describes the input of the forward(s) that filled `state` -- or BOTH, for a state whose first
 * B/2 captions were evaluated from token ids (inp_ids: int64 [B/2, L]) and whose last B/2 from soft rows (inp_soft: act
 * [B/2*L, V]): the step's D(real) and D(fake) passes written into the two halves of one state, differentiated in one call
 * (grads required, d_inp must be NULL). */
int gic_disc_bwd(const gic_disc_dims* dims, const gic_disc_params* params, const gic_disc_shadow* shadow,
                 const gic_disc_state* state, const gic_disc_bwd_ws* ws, const void* inp_soft, int64_t ld_inp,
                 const int64_t* inp_ids, int train, const float* d_logits, const gic_disc_grads* grads,
                 int accumulate, void* d_inp, int64_t ld_dinp, void* stream);

/* ------------------------------------------------------------------------------------------
 * Encoder (src/generator.py:8-25): ResNet trunk forward (frozen, BatchNorm on batch statistics) and the trainable
 * Linear + BatchNorm1d(momentum=0.01) head.  Activations are NHWC in the compute dtype ("act").
 */
/* NCHW f32 [N,3,S,S] -> zero-bordered NHWC4 act [N, S+2*pad, Wp, 4] (channel 3 = 0); Wp >= S+2*pad. */
int gic_pack_image(const float* nchw, void* out, int dtype, int N, int S, int pad, int Wp, void* stream);
/* conv weight [Cout,Cin,KH,KW] f32 -> act [Cout,KH,KW_pad,Cin_pad] (zero padded). */
int gic_repack_conv_weight(const float* w, void* out, int dtype, int Cout, int Cin, int KH, int KW, int Cin_pad, int KW_pad,
                           void* stream);
/* Implicit-GEMM convolution on MFMA: in act [N,H,W,Cin], w act [Cout,KH,KW,Cin], out act [N,Ho,Wo,Cout].  stats (optional,
 * f32 [stats_nrep][2*Cout], accumulated with atomics: the caller zeroes it; workgroup b adds into replica b % stats_nrep so that
 * few adders share an address, readers sum the replicas): per-channel sum and sum of squares of the f32 results = the batch
 * statistics nn.BatchNorm2d needs in train mode.  Replaces nn.Conv2d of the torchvision trunk (generator.py:12-14,22). */
int gic_conv2d(const void* in, const void* w, void* out, float* stats, int stats_nrep, int dtype, int N, int H, int W, int Cin,
               int Cout, int KH, int KW, int stride, int pad, void* stream);

/* Convolution whose INPUT is normalised on the fly: out = conv(pad(relu(gamma (in - mean) / sqrt(var + 1e-5) + beta))) with
 * mean / var from `in_stats` ([in_nrep][2*Cin] sums written by the gic_conv2d that produced `in`, over in_count rows).  Replaces the
 * bn -> ReLU -> conv links inside the torchvision residual blocks (src/generator.py:12-14) without the separate gic_bn_act pass
 * and without the normalised tensor: 1x1 windows rewrite each A tile in LDS before its MFMAs; 3x3 / stride 1 / pad 1 windows with
 * Cin % 64 == 0 keep the input patch of a 128-pixel tile in LDS for all nine taps and normalise it once per 64-channel chunk
 * (padding stays zero).  `stats` receives this convolution's own column sums as gic_conv2d does.  Returns GIC_STATUS_UNSUPPORTED
 * (and launches nothing) in f32 mode, for Cin > 1024 or Cin % 8 != 0, for any other window, or for shapes the kernels do not
 * take: the caller then runs gic_bn_act + gic_conv2d. */
int gic_conv2d_bn_in(const void* in, const float* in_stats, int in_nrep, const float* in_gamma, const float* in_beta, float in_count,
                     const void* w, void* out, float* stats, int stats_nrep, int dtype, int N, int H, int W, int Cin, int Cout, int KH,
                     int KW, int stride, int pad, void* stream);
/* The 1x1 convolution that opens a bottleneck block, with the PREVIOUS block's output formed on load: its A operand is
 *   x = relu( bn(in) + r ),   r = res (identity shortcut, res_stats == NULL) or bn_res(res) (projection shortcut),
 * with both BatchNorms on batch statistics (sums over `count` rows: in_stats [in_nrep][2*Cin], res_stats [res_nrep][2*Cin]) --
 * torchvision's `out = relu(bn3(conv3(..)) + identity)` followed by the next block's conv1 (src/generator.py:12-14) without the
 * separate normalise + add + relu pass: the workgroups of the first output-channel tile also write x to block_out [N,H,W,Cin]
 * (the next shortcut / the projection convolution read it from there).  in, res: act [N,H,W,Cin]; w act [Cout,Cin]; out act
 * [N,H,W,Cout]; stats as gic_conv2d.  GIC_STATUS_UNSUPPORTED (nothing launched) in f32 mode, for Cin % 8 != 0 or Cin > 2048, or for
 * shapes the 8-wave kernel does not take: the caller then runs gic_bn_act + gic_conv2d. */
int gic_conv1x1_res_in(const void* in, const float* in_stats, int in_nrep, const float* in_gamma, const float* in_beta, const void* res,
                       const float* res_stats, int res_nrep, const float* res_gamma, const float* res_beta, float count, void* block_out,
                       const void* w, void* out, float* stats, int stats_nrep, int dtype, int N, int H, int W, int Cin, int Cout,
                       void* stream);

/* (ABI v4) The tail of a bottleneck block and the head of the next one in ONE launch, with conv3's output never written
 * (src/generator.py:12-14: `out = relu(bn3(conv3(relu(bn2(y2)))) + shortcut)` followed by the next block's conv1):
 *   gic_conv1x1_bn_in_stats   conv3's BatchNorm column sums ONLY (its own statistics have to exist before any of its output can be
 *                             normalised): in = y2 act [rows, Cin] read as relu(bn(in)) as in gic_conv2d_bn_in, w act [Cout, Cin],
 *                             stats [stats_nrep][2*Cout] added to as gic_conv2d does; nothing else is written.
 *   gic_conv_b2b              recomputes conv3 from y2 (C2 channels; w3 act [4*C2, C2]), forms out = relu(bn3(.) + r), r = res (identity
 *                             shortcut, res_stats == NULL) or bn_res(res) (projection), writes it to block_out act [rows, 4*C2] and feeds
 *                             it to the next conv1 (w1n act [C1N, 4*C2]): y1n act [rows, C1N] and its column sums into stats1.
 *                             All BatchNorms on batch statistics over `count` rows.
 * Both return GIC_STATUS_UNSUPPORTED (nothing launched) in f32 mode and for shapes they have no kernel for (C2 in {64, 128},
 * C1N in {64, 128}, rows % 64 == 0, enough rows for the streaming kernel): the caller runs gic_conv2d_bn_in + gic_conv1x1_res_in. */
int gic_conv1x1_bn_in_stats(const void* in, const float* in_stats, int in_nrep, const float* in_gamma, const float* in_beta, float in_count,
                            const void* w, float* stats, int stats_nrep, int dtype, int64_t rows, int Cin, int Cout, void* stream);
int gic_conv_b2b(const void* y2, const float* stats2, int nrep2, const float* gamma2, const float* beta2, const void* w3, const float* stats3,
                 int nrep3, const float* gamma3, const float* beta3, const void* res, const float* res_stats, int res_nrep,
                 const float* res_gamma, const float* res_beta, float count, void* block_out, const void* w1n, void* y1n, float* stats1,
                 int nrep1, int dtype, int64_t rows, int C2, int C1N, void* stream);
/* out = [relu]( bn(y) + (res ? bn_res(res) : 0) ) over rows x C.  A BatchNorm takes its mean/var from `stats` (raw sums over
 * `count` rows; train mode) or from run_mean/run_var (eval mode); res_gamma == NULL -> the residual is added as is. */
int gic_bn_act(const void* y, const float* stats, const float* gamma, const float* beta, const float* run_mean,
               const float* run_var, const void* res, const float* res_stats, const float* res_gamma, const float* res_beta,
               const float* res_run_mean, const float* res_run_var, int stats_nrep, float count, int relu, void* out, int dtype,
               int64_t rows, int C, void* stream);
/* Stem: relu(bn(y)) then 3x3 / stride 2 / pad 1 max-pool.  y act [N,H,W,C] -> out act [N,(H+1)/2,(W+1)/2,C]. */
int gic_bn_relu_maxpool(const void* y, const float* stats, const float* gamma, const float* beta, const float* run_mean,
                        const float* run_var, int stats_nrep, float count, void* out, int dtype, int N, int H, int W, int C,
                        void* stream);
/* Global average pool: x act [N,HW,C] -> out act [N,C]. */
int gic_avgpool(const void* x, void* out, int dtype, int N, int HW, int C, void* stream);
/* Running mean/var of every trunk BatchNorm2d in one launch; `table_dev` is a DEVICE array built once by the caller. */
typedef struct gic_bn_running_desc {
  const float* stats;      /* [nrep][2C] raw sums of this step */
  float* running_mean;     /* [C] */
  float* running_var;      /* [C] */
  float count;             /* rows the sums were taken over */
  float momentum;
  int32_t C;
  int32_t nrep;
} gic_bn_running_desc;
int gic_bn_running_update(const gic_bn_running_desc* table_dev, int nlayers, void* stream);
/* nn.BatchNorm1d over the batch axis of x [B,E] (generator.py:16,24) forward / backward. */
int gic_bn1d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int training,
                 float momentum, float eps, float* y, float* xhat, float* invstd, int B, int E, void* stream);
int gic_bn1d_bwd(const float* dy, const float* xhat, const float* invstd, const float* gamma, int training, float* dx,
                 float* dgamma, float* dbeta, int B, int E, void* stream);
/* out[c] (+)= sum_r A[r*lda + c] (bias gradients). */
int gic_colsum(const void* A, int dtype, int64_t lda, int64_t rows, int64_t cols, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * get_losses (src/utils.py:10-53): losses[0]=g_loss, losses[1]=d_loss (device scalars) and, when the
 * d_* pointers are non-NULL, the gradients of d_loss w.r.t. (d_real, d_fake) and of g_loss w.r.t.
 * (g_out, and for rsgan d_real/d_fake through dg_real/dg_fake).
 */
int gic_gan_losses(int loss_type, const float* d_real, const float* d_fake, const float* g_out, int64_t n,
                   float* losses, float* dd_real, float* dd_fake, float* dg_out, float* dg_real, float* dg_fake,
                   void* stream);

/* CrossEntropyLoss over all rows (training.py:81-83): loss = device f32[1+rows] (loss[0] = mean, rest = per-row
 * scratch) and d_logits = (softmax - onehot)/rows (optional).  A target outside [0, V) makes loss[0] NaN (the reference's
 * nn.CrossEntropyLoss raises).  row_weight (optional, f32 [rows]): the policy-gradient form, loss = mean_r w_r * nll_r and
 * d_logits scaled by w_r -- the REINFORCE generator loss of the SeqGAN update with w = the roll-out rewards. */
int gic_xent(const void* logits, int dtype, int64_t rows, int32_t V, const int64_t* targets, float* loss,
             void* d_logits, const float* row_weight, void* stream);

/* SeqGAN Monte-Carlo rewards (BASELINE config 5; no reference counterpart): mc_logits f32 [(L-1), N, B, R] = D's logits on the
 * N roll-outs of every prefix length 1..L-1 (caption b, representation r), full_logits f32 [B, R] = D on the complete captions.
 * rewards f32 [B, L]: reward[b, t] = mean_{n,r} sigmoid(mc_logits[t, n, b, r]) for t < L-1, mean_r sigmoid(full_logits[b, r]) for
 * t = L-1. */
int gic_rollout_rewards(const float* mc_logits, const float* full_logits, float* rewards, int B, int L, int N, int R, void* stream);

/* ------------------------------------------------------------------------------------------
 * optimize(): clip_grad_norm_ + Adam (src/training.py:194-199, :24-26) over a flat f32 parameter arena.
 * step_count: device int64 (incremented here); norm_out: device f32 (pre-clip global L2 norm);
 * partials: device f32 scratch [gic_clip_adam_partials(n)].  Hyper-parameters are doubles: torch.optim.Adam forms
 * 1-beta, lr/(1-beta1^t) and sqrt(1-beta2^t) in Python float64 before touching f32 tensors, and so does the kernel.
 */
int64_t gic_clip_adam_partials(int64_t n);
int gic_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, double clip_norm, int64_t* step_count, float* norm_out,
                  float* partials, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GICAP_H_ */
