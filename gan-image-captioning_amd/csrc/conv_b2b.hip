// conv3 -> bn3 + shortcut + ReLU -> the next block's conv1, BACK TO BACK in one launch (round 3): the tail of a torchvision bottleneck
// and the head of the next one (reference src/generator.py:12-14) on the large maps, where both are HBM-bound.
//
//   y3   = conv3(relu(bn2(y2)))                     1x1, C2 -> C3 = 4 C2   (y2 = the raw output of conv2, normalised on load)
//   out  = relu(bn3(y3) + shortcut)                 the block output (written: the next block's shortcut)
//   y1n  = conv1_next(out)                          1x1, C3 -> C1N          (+ its BatchNorm column sums)
//
// As separate launches (conv1x1_stream for conv3, tile8's residual-on-load form for conv1_next) y3 is written and read again:
// 2 x 103 MB per block at 56 x 56 and batch 64, the largest tensor of the block, between two kernels that already run at 4-6 TB/s.
// Here y3 never reaches memory.  BatchNorm needs y3's batch statistics BEFORE any of it can be normalised, so conv3 runs twice:
// a statistics-only pass first (the streaming kernel without its stores: reads y2, 26 MB) and the recomputation in this kernel.
//
// The intermediate stays in REGISTERS.  A wave owns 16 pixels.  Per 64-channel chunk of C3 it computes conv3 in the transposed form
// (MFMA A operand = W3 rows, B operand = its pixels): a lane then holds ONE pixel and, with the chunk's W3 rows laid out in LDS in
// the order row x of block j <-> channel 16 (x >> 2) + 4 j + (x & 3), 16 CONSECUTIVE channels of it -- 32 contiguous bytes of the
// shortcut to load, of the block output to store, and, rounded to bf16, exactly two B-operand fragments of the second product
// (k slot (lg, e) <-> channel 16 lg + 8 ks + e; the W1n chunk is read with the same k order: one 16-byte LDS read per fragment).
// No LDS round trip and no workgroup barrier between the two products; conv1_next's output comes out transposed the same way
// (16 consecutive output channels of a pixel per lane: 32-byte stores), its column sums stay per lane in registers across all tiles.
// Only the weight chunks are shared: a two-stage LDS-DMA ring (W3 chunk + W1n chunk, both L2-resident), one barrier per chunk; the
// workgroup is persistent and small (weights ring + coefficient tables), so two or three share a CU and cover each other's latency.
#include <stdlib.h>

#include "conv_b2b.h"
#include "bn_fold.h"

namespace gic {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte accesses through a buffer descriptor: per-lane byte offset + scalar byte offset
__device__ __forceinline__ u32x4 buf_load16(const __amdgpu_buffer_rsrc_t r, const int voff, const int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store16(const u32x4 v, const __amdgpu_buffer_rsrc_t r, const int voff, const int soff) {
  // The tile offset rides in the per-lane offset, not in the scalar one: the compiler (hipcc 7.2) assumes a store of more than 8 bytes
  // with an SGPR offset needs no wait state before a VALU instruction overwrites its data registers and schedules one right behind
  // it; on gfx950 that instruction's result reached memory in place of the first dword (sporadically, lanes 12-15 of each row of 16).
#ifdef GIC_STORE_SOFF                                                      // (measurement build: the form that exposes the hazard)
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
#else
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, 0);
#endif
}

// Lane l's value of its neighbour l ^ X inside its row of 16 lanes, on the VALU (DPP: fused into the addition that consumes it).
// __shfl_xor is ds_bpermute_b32 -- an LDS instruction: 30 of them per reduce-scatter pair in four dependent stages, queued behind the
// weight fragment reads of all eight waves.
template <int CTRL>
__device__ __forceinline__ float row_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_xor8(const float v) { return row_dpp<0x128>(v); }                  // row_ror:8
__device__ __forceinline__ float row_xor4(const float v) {                                              // lanes with bit 2 clear read l + 4 (row_ror:12,
  const int x = __builtin_bit_cast(int, v);                                                             // banks 0 and 2), the others l - 4 (row_ror:4)
  int a = __builtin_amdgcn_update_dpp(0, x, 0x12C, 0xF, 0x5, false);
  a = __builtin_amdgcn_update_dpp(a, x, 0x124, 0xF, 0xA, false);
  return __builtin_bit_cast(float, a);
}
__device__ __forceinline__ float row_xor2(const float v) { return row_dpp<0x4E>(v); }                   // quad_perm:[2,3,0,1]
__device__ __forceinline__ float row_xor1(const float v) { return row_dpp<0xB1>(v); }                   // quad_perm:[1,0,3,2]

// Sum over the 16 lanes of a row (lr) of 16 per-lane values, value e ending up in lane lr == e: a reduce-scatter butterfly, 15 lane
// exchanges and 15 additions instead of 16 separate registers that live across the whole kernel.
__device__ __forceinline__ float row_reduce_scatter16(const float (&v)[16], const int lr) {
  float t[8], u[4], x[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const bool up = lr & 8; t[i] = (up ? v[i + 8] : v[i]) + row_xor8(up ? v[i] : v[i + 8]); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { const bool up = lr & 4; u[i] = (up ? t[i + 4] : t[i]) + row_xor4(up ? t[i] : t[i + 4]); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const bool up = lr & 2; x[i] = (up ? u[i + 2] : u[i]) + row_xor2(up ? u[i] : u[i + 2]); }
  const bool up = lr & 1;
  return (up ? x[1] : x[0]) + row_xor1(up ? x[0] : x[1]);
}

// C2: channels of y2 (64 | 128); C1N: output channels of the next conv1 (64 | 128); IDENT: identity shortcut (no BatchNorm of its own)
template <int C2, int C1N, bool IDENT>
__global__ __launch_bounds__(512) void conv_b2b_kernel(const B2bDesc d) {
  // RES (C2 == 64: the 56 x 56 boundaries): ALL weight chunks resident in LDS (64 / 96 KB), loaded once per persistent workgroup of eight
  // waves -- no DMA, no wait and no barrier inside the loop, the waves run free.  Otherwise (C2 == 128: 256 KB of weights) a THREE-stage
  // ring with one barrier per chunk: the weight pieces of chunk c + 2 are issued during chunk c.  (Loads return in order: with a
  // two-stage ring the L2-resident weight pieces of the next chunk sat in the queue behind the shortcut loads issued just before
  // them, which come from HBM, and every chunk's wait for its weights inherited a memory round trip.)
  constexpr bool RES = C2 == 64;
  constexpr int NSTG = 3;
  constexpr int NT = 512, NW = NT / 64, BM = 16 * NW, C3 = 4 * C2, NC = C3 / 64, KS1 = C2 / 32, G2 = C1N / 64;
  constexpr int W3_BYTES = 64 * C2 * 2, W1_BYTES = C1N * 128, STAGE = W3_BYTES + W1_BYTES;
  constexpr int PW3 = W3_BYTES / 16 / NT, PW1 = W1_BYTES / 16 / NT;       // DMA pieces per thread and chunk
  constexpr int COEF0 = (RES ? NC : NSTG) * STAGE;
  constexpr int PW = PW3 + PW1;
  constexpr int ROW3 = C2 * 2, CH3 = ROW3 / 16;                           // bytes / 16-byte pieces of a W3 row (128 | 256 B)
  static_assert(PW3 >= 1 && PW1 >= 1, "piece counts");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* coef2 = (float*)(smem + COEF0);                                 // [C2][2]  bn2 scale, shift
  float* coef3 = coef2 + 2 * C2;                                         // [C3][2]  bn3
  float* coefr = coef3 + 2 * C3;                                         // [C3][2]  shortcut (identity: 1, 0)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int ntiles = d.M / BM;

  const __amdgpu_buffer_rsrc_t rsW3 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w3, 0, C3 * C2 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w1n, 0, C1N * C3 * 2, 0x00020000);
  const int wbase = (tid & ~63) * 16;
  // activations through buffer descriptors too: a 32-bit byte offset per lane and tile, the chunk's offset as the instruction's scalar
  // offset -- no 64-bit address arithmetic per access (the kernel is bound by instruction issue, not by memory)
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)d.y2, 0, (int)d.y2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)d.res, 0, (int)d.res_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, (int)d.res_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(d.y1n, 0, (int)d.y1n_bytes, 0x00020000);

  // ---- weight chunk c -> ring stage st_.  LDS images: W3 chunk [64 rows][C2] with LDS row L = 16 j + x holding channel
  //      16 (x >> 2) + 4 j + (x & 3) of the chunk; W1n chunk [C1N rows][64] with LDS row 64 g + 16 jb + x holding output channel
  //      64 g + 16 (x >> 2) + 4 jb + (x & 3); 16-byte pieces XOR-swizzled by (row >> 1) & 7 within each 128 bytes of a row.
  static_assert(PW3 <= 4 && PW1 <= 4, "piece offset arrays");
  int w3_off[4], w1_off[4];                                              // byte offsets of this thread's pieces inside chunk 0 (literal bounds:
                                                                         // with [PW3] the host pass silently drops the kernel's instantiation)
#pragma unroll
  for (int i = 0; i < PW3; ++i) {
    const int q = tid + NT * i, L = q / CH3, slot = q % CH3;
    const int j = L >> 4, x = L & 15, ch = 16 * (x >> 2) + 4 * j + (x & 3);
    const int kp = (slot & ~7) | ((slot & 7) ^ ((L >> 1) & 7));           // logical 16-byte piece of the row
    w3_off[i] = (ch * C2 + kp * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < PW1; ++i) {
    const int q = tid + NT * i, L = q >> 3, slot = q & 7;
    const int g = L >> 6, jb = (L >> 4) & 3, x = L & 15, qo = 64 * g + 16 * (x >> 2) + 4 * jb + (x & 3);
    const int kp = slot ^ ((L >> 1) & 7);
    w1_off[i] = (qo * C3 + kp * 8) * 2;
  }
  auto issue_w_ = [&](const int c, const int st_, const int (&w3o)[4], const int (&w1o)[4]) {
    unsigned char* s0 = smem + st_ * STAGE;
    // chunk c: W3 rows 64 c .. (64 C2 elements further), W1n columns 64 c ..; c == NC: nothing (zero fill, no traffic)
    // (the scalar offset is not part of the descriptor's range check: the out-of-range marker goes into the per-lane offset)
    const bool ok = c < NC;
    const int so3 = ok ? c * (64 * C2 * 2) : 0, so1 = ok ? c * 128 : 0;
    const int oob = ok ? 0 : (int)0x80000000;                            // OR-ed into the per-lane offset: past every extent
#pragma unroll
    for (int i = 0; i < PW3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW3, (lds_void_ptr)(s0 + i * (NT * 16) + wbase), 16, w3o[i] | oob, so3, 0, 0);
#pragma unroll
    for (int i = 0; i < PW1; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_void_ptr)(s0 + W3_BYTES + i * (NT * 16) + wbase), 16, w1o[i] | oob, so1, 0, 0);
  };

  auto issue_w = [&](const int c, const int st_) { issue_w_(c, st_, w3_off, w1_off); };
  if constexpr (RES) {
#pragma unroll
    for (int c = 0; c < NC; ++c) issue_w(c, c);
  } else {
    issue_w(0, 0);
    issue_w(1, 1);
  }
  // ---- coefficient tables
  for (int c = tid; c < C2 + 2 * C3; c += NT) {
    float sc, sh;
    if (c < C2 + C3) {
      const bool two = c < C2;
      const int cc = two ? c : c - C2, Cn = two ? C2 : C3;
      float s1, s2;
      fold_replicas(two ? d.stats2 : d.stats3, two ? d.nrep2 : d.nrep3, Cn, cc, s1, s2);
      const float mean = s1 * d.inv_count, var = fmaxf(s2 * d.inv_count - mean * mean, 0.f);
      sc = (two ? d.gamma2 : d.gamma3)[cc] * rsqrtf(var + 1e-5f);        // kBnEps of encoder.hip (nn.BatchNorm2d default)
      sh = (two ? d.beta2 : d.beta3)[cc] - mean * sc;
    } else {
      const int cc = c - C2 - C3;
      sc = 1.f; sh = 0.f;                                                // identity shortcut
      if (d.res_stats) {                                                 // projection shortcut: its own BatchNorm
        float s1, s2;
        fold_replicas(d.res_stats, d.res_nrep, C3, cc, s1, s2);
        const float mean = s1 * d.inv_count, var = fmaxf(s2 * d.inv_count - mean * mean, 0.f);
        sc = d.res_gamma[cc] * rsqrtf(var + 1e-5f);
        sh = d.res_beta[cc] - mean * sc;
      }
    }
    coef2[2 * c] = sc; coef2[2 * c + 1] = sh;                            // (the three tables are contiguous)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the table writes are done before this thread reaches the first barrier
  if constexpr (RES) {                                                   // ... and its weight pieces have landed: the one barrier of the resident form
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  float st_s[G2], st_q[G2];                                              // conv1_next's column sums over this wave's pixels, all tiles: lane (lr, lg) holds column 64 g + 16 lg + lr
#pragma unroll
  for (int g = 0; g < G2; ++g) st_s[g] = st_q[g] = 0.f;

  // ---- loads run AHEAD: the shortcut's 32 bytes of a pixel two chunks ahead (two register pairs, a pair is refilled right after its use,
  //      across tile boundaries), the next tile's y2 fragments during the current tile's last chunk.  With one chunk of cover every chunk
  //      waited out a full memory round trip: SQ_WAIT_ANY 59 % of the waves' cycles (tools/trunk_pmc.sh).
  auto offs = [&](const int tile, int& yoff, int& roff, int& noff) {
    // byte offsets of this lane's pixel (M is a multiple of 64 and every tensor is below 2 GiB: the launch code); past the last tile:
    // the out-of-range marker (loads return zero without traffic, the count of outstanding operations stays uniform)
    const int m = tile * BM + w * 16 + lr;
    const int oob = tile < ntiles ? 0 : (int)0x80000000;
    yoff = ((m * C2 + lg * 8) * 2) | oob; roff = ((m * C3 + lg * 16) * 2) | oob; noff = ((m * C1N + lg * 16) * 2) | oob;
  };
  int yoff, roff, noff;
  offs(blockIdx.x, yoff, roff, noff);
  bf16x8 fy[KS1], fyn[KS1];
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks) fyn[ks] = __builtin_bit_cast(bf16x8, buf_load16(rsY, yoff, ks * 64));
  u32x4 ra0 = buf_load16(rsR, roff, 0), ra1 = buf_load16(rsR, roff + 16, 0);          // chunk 0 (even chunks: pair a)
  u32x4 rb0 = buf_load16(rsR, roff, 128), rb1 = buf_load16(rsR, roff + 16, 128);      // chunk 1 (odd chunks: pair b)

  int st = 0;
  bool first = true;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int yoff_n, roff_n, noff_n;
    offs(tile + gridDim.x, yoff_n, roff_n, noff_n);
    f32x4 acc2[G2][4];
#pragma unroll
    for (int g = 0; g < G2; ++g)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) acc2[g][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](const int c, u32x4& r0, u32x4& r1) {
      // This thread's weight pieces of chunk c have landed once only what was issued BEHIND them is outstanding (one in-order counter for
      // loads, LDS-DMA and stores): the previous chunk's two block-output stores and two shortcut loads -- behind a tile's last chunk also
      // its y1n stores and the next tile's y2 loads.  Nothing waits for a store to complete.
      // (the workgroup's very first chunk: only the prologue's y2 and shortcut loads are behind its weight pieces)
      if constexpr (!RES) {
        // Behind chunk c's weight pieces (issued two chunks ago): per chunk 2 stores + 2 shortcut loads, the next chunk's PW pieces; around
        // a tile boundary also the y1n stores and the next tile's y2 loads; in the workgroup's first two chunks the prologue's loads.
        if (first && c == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + KS1 + 4) : "memory");
        else if (first && c == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS1 + 8 + PW) : "memory");
        else if (c < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + PW + KS1 + 2 * G2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + PW) : "memory");
        __builtin_amdgcn_s_barrier();                                    // chunk c's stage (and, the first time, the tables) visible; the stage of chunk c - 1 is free
        asm volatile("" ::: "memory");
        {                                                                // chunk c + 2 (of the next tile; none behind the last tile)
          const bool more = tile + (int)gridDim.x < ntiles;
          const int cn = c + 2 < NC ? c + 2 : (more ? c + 2 - NC : NC);
          issue_w(cn, st >= 1 ? st - 1 : NSTG - 1);                      // (st + 2) % 3
        }
      }
      if (c == 0) {
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const float4* cp = (const float4*)(coef2 + 2 * (ks * 32 + lg * 8));
          const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
          const float scl[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
          const float sft[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) fy[ks][e] = (bf16_t)fmaxf((float)fyn[ks][e] * scl[e] + sft[e], 0.f);
        }
      }
      const unsigned char* sW3 = smem + (RES ? c : st) * STAGE;
      const unsigned char* sW1 = sW3 + W3_BYTES;
      // ---- conv3, transposed: block j, MFMA row x <-> channel 16 (x >> 2) + 4 j + (x & 3): lane (lr, lg) ends up with channels 16 lg + 4 j + r
      f32x4 acc1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int L = 16 * j + lr, kp = ks * 4 + lg;
          const bf16x8 fw = *(const bf16x8*)(sW3 + L * ROW3 + (((kp & ~7) | ((kp & 7) ^ ((L >> 1) & 7))) << 4));
          acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fy[ks], acc1[j], 0, 0, 0);
        }
      }
      // ---- bn3 + shortcut + ReLU: 16 consecutive channels 64 c + 16 lg + e, e = 4 j + r
      const bf16_t* rv0 = (const bf16_t*)&r0;
      const bf16_t* rv1 = (const bf16_t*)&r1;
      bf16x8 a2[2];                                                      // the block output's 16 channels = two B-operand fragments of conv1_next
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const float4 p3 = *(const float4*)(coef3 + 2 * (c * 64 + lg * 16 + e));     // (sc, sh) of channels e, e + 1
        float x0 = (float)(e < 8 ? rv0[e] : rv1[e - 8]), x1 = (float)(e + 1 < 8 ? rv0[e + 1] : rv1[e + 1 - 8]);
        if constexpr (!IDENT) {
          const float4 pr = *(const float4*)(coefr + 2 * (c * 64 + lg * 16 + e));
          x0 = x0 * pr.x + pr.y; x1 = x1 * pr.z + pr.w;
        }
        const float v0 = fmaxf(acc1[e >> 2][e & 3] * p3.x + p3.y + x0, 0.f);
        const float v1 = fmaxf(acc1[(e + 1) >> 2][(e + 1) & 3] * p3.z + p3.w + x1, 0.f);
        a2[e >> 3][e & 7] = (bf16_t)v0;
        a2[(e + 1) >> 3][(e + 1) & 7] = (bf16_t)v1;
      }
      if (c == NC - 1) {                                                 // the next tile's y2 fragments (behind this chunk's weight pieces)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) fyn[ks] = __builtin_bit_cast(bf16x8, buf_load16(rsY, yoff_n, ks * 64));
      }
      buf_store16(__builtin_bit_cast(u32x4, a2[0]), rsO, roff, c * 128);
      buf_store16(__builtin_bit_cast(u32x4, a2[1]), rsO, roff + 16, c * 128);
      // refill this pair: the shortcut of chunk c + 2 (of the next tile behind the last two chunks)
      if (c + 2 < NC) { r0 = buf_load16(rsR, roff, (c + 2) * 128); r1 = buf_load16(rsR, roff + 16, (c + 2) * 128); }
      else { r0 = buf_load16(rsR, roff_n, (c + 2 - NC) * 128); r1 = buf_load16(rsR, roff_n + 16, (c + 2 - NC) * 128); }
      // ---- conv1_next, transposed: acc2[g][jb] += W1n rows (A operand) x a2 (B operand, k slot (lg, e) <-> chunk channel 16 lg + 8 ks + e)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int g = 0; g < G2; ++g)
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) {
            const int L = 64 * g + 16 * jb + lr, kp = lg * 2 + ks;       // 16-byte piece of the row: channels 16 lg + 8 ks .. + 7
            const bf16x8 fw = *(const bf16x8*)(sW1 + L * 128 + ((kp ^ ((L >> 1) & 7)) << 4));
            acc2[g][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, a2[ks], acc2[g][jb], 0, 0, 0);
          }
      }
      st = st == NSTG - 1 ? 0 : st + 1;
    };
#pragma unroll 1
    for (int c = 0; c < NC; c += 2) {
      chunk(c, ra0, ra1);
      chunk(c + 1, rb0, rb1);
    }
    // ---- y1n: lane (lr, lg) holds output channels 64 g + 16 lg + 4 jb + r of its pixel: 32 contiguous bytes per g
#pragma unroll
    for (int g = 0; g < G2; ++g) {
      bf16x8 o[2];
      float vs[16], vq[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc2[g][e >> 2][e & 3];
        o[e >> 3][e & 7] = (bf16_t)v;
        vs[e] = v;
        vq[e] = v * v;
      }
      st_s[g] += row_reduce_scatter16(vs, lr);
      st_q[g] += row_reduce_scatter16(vq, lr);
      buf_store16(__builtin_bit_cast(u32x4, o[0]), rsN, noff, g * 128);
      buf_store16(__builtin_bit_cast(u32x4, o[1]), rsN, noff + 16, g * 128);
    }
    yoff = yoff_n; roff = roff_n; noff = noff_n;
    first = false;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // a trailing zero-fill DMA must land before the LDS is handed on

  // ---- conv1_next's BatchNorm column sums: fold the 16 pixels of the wave (lanes lr), then the eight waves through LDS (the weight ring is
  //      idle now): ONE atomic per column and workgroup (one per wave measured 250-460 us per launch: half a million atomics on a few
  //      hundred addresses)
  __syncthreads();
  float* red = (float*)smem;                                             // [NW waves][2][C1N]
#pragma unroll
  for (int g = 0; g < G2; ++g) {
    const int col = g * 64 + lg * 16 + lr;
    red[(w * 2) * C1N + col] = st_s[g];
    red[(w * 2 + 1) * C1N + col] = st_q[g];
  }
  __syncthreads();
  for (int t_ = tid; t_ < 2 * C1N; t_ += NT) {
    const int tid = t_;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[ww * 2 * C1N + tid];
    atomicAdd(d.stats1 + (long)(blockIdx.x % d.nrep1) * 2 * C1N + tid, v);
  }
}

template <int C2, int C1N, bool IDENT>
bool launch_b2b(const B2bDesc& d, hipStream_t stream) {
  constexpr int C3 = 4 * C2;
  constexpr bool RES = C2 == 64;
  constexpr int NT = 512;
  constexpr size_t lds = (RES ? C3 / 64 : 3) * (size_t)(64 * C2 * 2 + C1N * 128) + (size_t)(C2 + 2 * C3) * 8;
  static_assert(lds <= 160 * 1024, "conv_b2b LDS budget");
  static LdsGrant granted;
  if (!grant_lds(conv_b2b_kernel<C2, C1N, IDENT>, lds, granted)) return false;
  // persistent workgroups per CU: as many as the LDS lets share a CU, two at most (measured: 3 and 4 lose)
  static const int per_cu = [] { const char* e = getenv("GIC_B2B_WG_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : (lds > 80 * 1024 ? 1 : 2); }();
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 256; }
    return n;
  }();
  const int tiles = d.M / (NT / 4);
  const int grid = tiles < cus * per_cu ? tiles : cus * per_cu;
  hipLaunchKernelGGL((conv_b2b_kernel<C2, C1N, IDENT>), dim3((unsigned)grid), dim3(NT), lds, stream, d);
  return true;
}

}  // namespace

bool try_conv_b2b(const B2bDesc& d, int C2, int C1N, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_CONV_B2B") != nullptr;
  if (off || d.M <= 0 || d.M % 128 || d.nrep1 < 1) return false;
  B2bDesc dd = d;
  dd.y1n_bytes = (unsigned)((long)d.M * C1N * 2);
  const long C3 = 4l * C2;
  if ((long)d.M * C3 * 2 >= (1l << 31)) return false;                    // 32-bit byte offsets into every activation
  for (const void* p : {d.y2, d.w3, d.res, d.w1n, (const void*)d.out, (const void*)d.y1n})
    if (!p || (((uintptr_t)p) & 15)) return false;
  const bool ident = d.res_stats == nullptr;
  if (C2 == 64 && C1N == 64) return ident ? launch_b2b<64, 64, true>(dd, stream) : launch_b2b<64, 64, false>(dd, stream);
  if (C2 == 64 && C1N == 128) return ident ? launch_b2b<64, 128, true>(dd, stream) : launch_b2b<64, 128, false>(dd, stream);
  if (C2 == 128 && C1N == 128) return ident ? launch_b2b<128, 128, true>(dd, stream) : launch_b2b<128, 128, false>(dd, stream);
  if (C2 == 128 && C1N == 256) return ident ? launch_b2b<128, 256, true>(dd, stream) : launch_b2b<128, 256, false>(dd, stream);
  return false;
}

}  // namespace gic
