// Fused per-timestep kernels of the caption roll-out (reference src/generator.py:60-76) -- launchers shared with decoder.hip.
//
// One decode step = TWO launches (the two all-to-all seams of the step: every hidden unit feeds every vocabulary logit, and
// every logit feeds the argmax that selects the next input):
//   lstm_step   [argmax of the previous step's partials -> next-input embedding gather] -> gates GEMM on MFMA -> LSTM pointwise
//   vocab_step  vocabulary GEMM on MFMA -> + bias + Gumbel(u) -> * temperature -> per-tile softmax partials (max, sum, argmax)
//               and the UNNORMALISED probabilities e = exp(y - tile max)
// and one launch after the last step (sample_finish): global max / sum per (caption, step) from the partials, token ids,
// probabilities p = e * exp(tile max - global max) / global sum.
// A grid-wide barrier inside one persistent launch costs 4.1-4.8 us on this chip (MI355X_MICROARCH.md price list, barrier-xcd; the
// single-launch roll-out built in round 2 measured ~19 us per barrier with the step's dirty lines to write back: 943 us per roll-out
// against 423 us; removed in round 3) against 1.45 us for a dependent kernel boundary, so the seams stay kernel
// boundaries and each kernel is sized to its floor:
// the 14 MB of decoder weights stay resident in the eight XCD L2s (4 MB each) between steps because the block -> weight-slice
// map is the same in every step.
#pragma once
#include "common.h"

namespace gic {

constexpr int kStepRows = 64;      // batch rows per block (4 MFMA tiles)
constexpr int kVocabTile = 64;     // vocabulary entries per vocab_step block
constexpr int kUnitsPerBlock = 4;  // hidden units per lstm_step block (4 gates x 4 units = one 16-wide MFMA tile)

struct LstmStepArgs {
  const void* xh_t = nullptr;        // act [B, ldx]: [x_t | h_{t-1}] (the x part is ignored when `gather`)
  void* xh_next = nullptr;           // act [B, ldx]: h_t is written at [:, din:]
  const void* wcat = nullptr;        // act [4H, ldx] = [w_ih | w_hh]
  const float* bsum = nullptr;       // [4H]
  const float* c_prev = nullptr;     // [B, H]
  float* c_new = nullptr;            // [B, H]
  float* gates = nullptr;            // [B, 4H] post-activation i,f,g,o (saved for backward) or null
  void* h_up = nullptr; long ld_up = 0;     // act: next layer's input slot, or null
  void* h_out = nullptr; long ld_out = 0;   // act: hout[b, t, :], or null
  int B = 0, H = 0, din = 0; long ldx = 0;
  // layer 0, t > 0: x_t = embed[id_{t-1}], id = argmax over the previous step's partials (or the forced trajectory)
  int gather = 0;
  int gw = 0;                        // width of the gathered x (= embedding dim): columns [0, gw) of the input; 0 = din
  const float* embed = nullptr; int V = 0;
  const unsigned long long* rowkey = nullptr;      // [B] argmax keys of step t-1 (vocab_step's atomicMax; see row_key())
  const int64_t* force_ids = nullptr; long force_stride = 0; const int32_t* force_len = nullptr; int tprev = 0;
  int dbg = 0;
};

struct VocabStepArgs {
  const void* h = nullptr; long ldh = 0;     // act rows: h + b*ldh, H values each
  const void* wout = nullptr;                // act [V, H]
  const float* bias = nullptr;               // [V]
  const float* u = nullptr;                  // explicit uniforms [B, V] of this step, or null -> Philox(seed, rng_stream)
  uint64_t seed = 0, rng_stream = 0;
  float temperature = 1.f;
  const float* t_dev = nullptr;              // non-null: temperature / seed are read from device memory (gic_step_scalars), the values
  const uint64_t* seed_dev = nullptr;        // above are ignored
  int pretrain = 0;
  void* out = nullptr; long out_stride = 0;  // act: out + b*out_stride + v (e or raw logits); null: ids only
  float* part_m = nullptr; float* part_s = nullptr; int nblk = 0;   // [B][nblk] per-tile max / sum of exp
  unsigned long long* rowkey = nullptr;      // [B], zeroed by the caller: atomicMax of (ordered tile max, ~first maximal index)
  int B = 0, V = 0, H = 0;
  int dbg = 0;
};

struct SampleFinishArgs {
  const float* part_m = nullptr; const float* part_s = nullptr;   // [L][B][nblk]
  const unsigned long long* rowkey = nullptr;                       // [L][B]
  int nblk = 0, B = 0, L = 0, V = 0, E = 0;
  int pretrain = 0;
  void* out = nullptr;                       // act [B, L, V] or null
  int64_t* ids = nullptr;                    // [B, L]
  const int64_t* force_ids = nullptr; const int32_t* force_len = nullptr;
  const float* embed = nullptr; void* xh0 = nullptr; long ldx0 = 0;     // x rows of XH_0 slots 1..L-1 (for the weight gradient) or null
};

// One BPTT step of one layer (reverse of lstm_step): dh = dh_above + dgates_{t+1} W_hh [+ dgates^{l+1}_t W_ih^{l+1}], then the
// cell's pointwise backward -> dgates_t, dc.  Tile = 16 batch rows x 16 hidden units per block, K (= 4H per segment) split over
// the 8 waves, operands straight from L2 (the k-contiguous operand of W is the transposed weight image wcat_t).
struct LstmBwdStepArgs {
  const float* dh_above = nullptr; long ld_above = 0;   // f32 rows (top layer: dhout[b, t, :]) or null
  const float* dh_extra = nullptr;                      // f32 [B, H] added as is (attention decoder: gradient through W_h h_{t-1}) or null
  int zero_extra = 0;                                   // != 0: dh_extra is zeroed behind the read (the next split-K product adds into it)
  const void* dg_next = nullptr;                        // act [B, 4H] dgates of step t+1, this layer (null at t = L-1)
  const void* w_rec = nullptr;                          // act rows j of wcat_t (+ din rows): [H][4H], W_hh^T
  const void* dg_up = nullptr;                          // act [B, 4H] dgates of step t, layer l+1 (null for the top layer)
  const void* w_up = nullptr;                           // act rows j of the upper layer's wcat_t: [H][4H], W_ih^{l+1,T}
  const float* gates = nullptr;                         // [B, 4H] post-activation i,f,g,o of step t
  const float* c_prev = nullptr; const float* c_cur = nullptr;   // [B, H]
  float* dc_state = nullptr;                            // [B, H] in/out
  void* dgates = nullptr;                               // act [B, 4H] out
  int B = 0, H = 0;
};
int lstm_bwd_step(const LstmBwdStepArgs& a, int dtype, hipStream_t stream);

// true if the fused kernels take these shapes (else the caller uses the generic GEMM + pointwise launches)
bool decoder_step_supported(int dtype, int V, int E, int H, int NL);
// most batch rows the fused roll-out takes (GIC_FUSED_ROLLOUT_MAX_ROWS, default 512): beyond it the per-step products are large GEMMs
int decoder_step_max_rows();
size_t decoder_step_part_floats(int B, int L, int V);       // floats in the partials scratch: [2][L][B][nblk] floats + [L][B] 64-bit keys + 2 reserved words
void decoder_step_debug(int v);                             // phase-ablation knob of tools/rollout_bench.py (0 = normal)
int lstm_step(const LstmStepArgs& a, int dtype, hipStream_t stream);
int vocab_step(const VocabStepArgs& a, int dtype, hipStream_t stream);
int sample_finish(const SampleFinishArgs& a, int dtype, hipStream_t stream);

}  // namespace gic
