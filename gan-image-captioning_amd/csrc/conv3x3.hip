// 3x3 / stride 1 / pad 1 convolution of the ResNet trunk (torchvision conv3x3 inside the residual blocks, reference
// src/generator.py:12-14) with the INPUT PATCH RESIDENT IN LDS, optional BatchNorm + ReLU of the input on load, bf16, NHWC,
// BatchNorm column sums in the epilogue.
//
// The implicit-GEMM kernel of gemm.hip (tile8) gathers the A operand tap by tap: every input pixel crosses L2 -> LDS nine
// times, one LDS-DMA piece per 16 bytes and tap, and the K loop's time is set by the DMA pieces a workgroup issues and lands per
// K step (measured, tools/conv_stamps.py with GIC_GEMM_DBG: 0.36 us per K step for the DMA alone, 0.44 us for LDS reads + MFMA,
// 0.60 us together).  Here a workgroup loads the zero-padded input patch of its 128 output pixels ONCE per 64-channel chunk
// (double-buffered across chunks) and reads the nine taps' MFMA fragments from it at shifted addresses; only the weight tile
// of a K step still streams through the ring (half the pieces of a 128x128 step).  With the patch in LDS the producer's
// BatchNorm + ReLU costs one pass over the patch instead of one per tap, so the separate bn_act launch that wrote the normalised
// tensor (and the tensor) disappears: gic_conv2d_bn_in takes 3x3 convolutions through this kernel.
//
// Geometry.  The zero-padded images stacked on top of each other form rows gp = n (H+2) + hp of W+2 pixels; output pixel
// (n, ho, wo) reads tap (r, s) at padded row n (H+2) + ho + r, column wo + s.  A tile of consecutive output pixels therefore
// reads a window of consecutive padded rows [gp0, gp1]: the patch, pixel index pp = (gp - gp0)(W+2) + column, 128 bytes
// (64 channels) per pixel and chunk, 16-byte pieces XOR-swizzled by (pp >> 1) & 7 on the global side of the DMA (LDS-DMA writes a
// wave's 64 pieces back to back) so that the 16 pixels of a fragment read fall on different banks.  Halo pixels are DMA'd from
// beyond the buffer descriptor's extent (zero fill, no traffic) and stay zero under BatchNorm-on-load (the reference pads the
// normalised tensor).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>

#include "conv3x3.h"
#include "bn_fold.h"

namespace gic {
#ifdef GIC_STAMPS
__device__ unsigned long long g_cstamp[2][8][2];     // [first | last workgroup][phase][shader clock, 100 MHz real time]
#define CSTAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) { \
  unsigned long long* p_ = g_cstamp[blockIdx.x == 0 ? 0 : 1][i]; p_[0] = __builtin_amdgcn_s_memtime(); p_[1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define CSTAMP(i) do {} while (0)
#endif
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Which of the 16 pixels of a fragment's segment MFMA row x holds.  ds_read_b128 serves a wave in groups of lanes {0-3, 12-15,
// 20-27}, ...: rows 0-3 and 12-15 of one 16-byte channel chunk together with rows 4-11 of the NEXT chunk.  With pixel p's chunk c
// stored at slot c ^ ((p >> 1) & 7) of its 128 bytes, such a group is conflict-free for every tap (every shift of the segment inside
// the patch) iff rows 4-11 hold the pixels of one parity and rows 0-3, 12-15 those of the other (consecutive rows measured 23-44 %
// of the LDS cycles as bank conflicts: the tap offset moves the segment off the alignment the XOR pattern assumes).
__device__ __forceinline__ int frag_row(int x) { return x < 4 ? 2 * x + 1 : (x < 12 ? 2 * (x - 4) : 2 * (x - 12) + 9); }

// a / b for 0 <= a < 2^23, b > 0 with inv = 1.f / b: a float product and one correction step instead of the ~40-instruction integer
// division (a workgroup needs a dozen of them before it can issue its first DMA: measured 1.2-1.8 us of its 6.6 us at 56x56)
__device__ __forceinline__ int fdiv(int a, int b, float inv) {
  int q = (int)((float)a * inv);
  const int r = a - q * b;
  q += (r >= b) - (r < 0);
  return q;
}

// Everything the kernel reads from its arguments, compact and in one struct: the scalar loads of a few adjacent cache lines leave in one
// batch at the top (fields of the 384-byte GemmDesc, fetched where first used, cost the prologue five serial round trips: ~1.2 us).
struct PatchDesc {
  const void* A; const void* B; void* C; float* stats;
  const float* in_stats; const float* in_gamma; const float* in_beta;
  int M, N, H, W, Cin, ldb, ldc, stats_nrep, in_nrep;
  float in_inv_count;
  int tiles_m;              // row tiles
  int tiles_n;              // > 0: output-channel tiles, and those of one row tile are neighbours on an XCD; 0: the row tiles of a channel tile are
  int tpi;                  // > 0: tiles never cross an image (tpi tiles per image, the last one short); 0: 128 consecutive rows of M
  int nchunks;              // Cin / 64
  unsigned a_bytes, b_bytes;
};

// BN output channels per workgroup (64 | 128); P = 16-byte patch pieces per thread and chunk (patch buffer = P * 8 KB = 64 P pixels);
// MULTI: more than one 64-channel chunk (double-buffered patch, the next chunk's pieces issued during the first P taps);
// ABN: BatchNorm + ReLU of the input on load.
template <int BN, int P, bool MULTI, bool ABN>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const PatchDesc d) {
  constexpr int BM = 128, NT = 512, NS = 3;
  constexpr int TM = 2, TN = BN / 32, CB = BN / 64;
  constexpr int PATCH = P * 8192, B_BYTES = BN * 128, NBUF = MULTI ? 2 : 1;
  constexpr int RING0 = NBUF * PATCH, COEF0 = RING0 + NS * B_BYTES;
  constexpr int SC = BN * 2 + 16;
  constexpr int EPI_BYTES = BM * SC + 8 * (BN / 2) * 2 * 4;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(COEF0 >= EPI_BYTES, "the C tile of the epilogue overlays the patch and the ring");
  static_assert(!MULTI || P <= 10 - NS, "the next chunk's patch pieces must be older than the weight tile of its first tap");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  CSTAMP(0);
  const int wr = w >> 1, wc = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int M = d.M, N = d.N, H = d.H, W = d.W, Cin = d.Cin;
  const PatchDesc& pa = d;
  const unsigned a_bytes = d.a_bytes, b_bytes = d.b_bytes;
  const int PH = H + 2, PW = W + 2, HW = H * W;

  const float iPW = 1.f / (float)PW, iPH = 1.f / (float)PH, iHW = 1.f / (float)HW, iW = 1.f / (float)W;
  // neighbours on one XCD (xcd_run): the output-channel tiles of one row tile (tiles_n > 0: the patch leaves HBM once) or, as tile8's
  // default, consecutive row tiles of one channel tile (the weights once)
  const int bid = xcd_run(blockIdx.x, gridDim.x);
  int tile_n, tile_m;
  if (pa.tiles_n > 0) {
    tile_m = fdiv(bid, pa.tiles_n, 1.f / (float)pa.tiles_n);
    tile_n = bid - tile_m * pa.tiles_n;
  } else {
    tile_n = fdiv(bid, pa.tiles_m, 1.f / (float)pa.tiles_m);
    tile_m = bid - tile_n * pa.tiles_m;
  }
  const int bn0 = tile_n * BN;
  int bm0, mlim;
  if (pa.tpi > 0) {
    const int n = fdiv(tile_m, pa.tpi, 1.f / (float)pa.tpi);
    bm0 = n * HW + (tile_m - n * pa.tpi) * BM;
    mlim = min(bm0 + BM, (n + 1) * HW);
  } else {
    bm0 = tile_m * BM;
    mlim = min(bm0 + BM, M);
  }
  // window of padded rows: first tap row of the first pixel .. last tap row of the last pixel
  const int n_first = fdiv(bm0, HW, iHW), ho_first = fdiv(bm0 - n_first * HW, W, iW);
  const int n_last = fdiv(mlim - 1, HW, iHW), ho_last = fdiv(mlim - 1 - n_last * HW, W, iW);
  const int gp0 = n_first * PH + ho_first;
  const int npix = (n_last * PH + ho_last + 2 - gp0 + 1) * PW;          // <= 64 P (launch code)

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, (int)b_bytes, 0x00020000);
  const int wbase = (tid & ~63) * 16;

  // ---- this thread's patch pieces: piece q = tid + 512 i = pixel q >> 3, physical slot q & 7 = logical 16-byte chunk ^ swizzle(pixel)
  int poff[P];              // element offset of (pixel, chunk 0 channels of the piece) in the input, or -1: zero (halo / past the patch)
  {
    // pixel of piece i = (tid >> 3) + 64 i: (slot, column) and (image, padded row) advance by constants, one division each up front
    const int c16 = (tid & 7) ^ ((tid >> 4) & 7);                       // (q & 7) ^ ((pp >> 1) & 7): the same for all of a thread's pieces
    const int dslot = fdiv(64, PW, iPW), dcol = 64 - dslot * PW;
    int pp = tid >> 3;
    int slot = fdiv(pp, PW, iPW), col = pp - slot * PW;
    int n = fdiv(gp0 + slot, PH, iPH), hp = gp0 + slot - n * PH;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const bool ok = pp < npix && hp >= 1 && hp <= H && col >= 1 && col <= W;
      poff[i] = ok ? ((n * H + hp - 1) * W + col - 1) * Cin + c16 * 8 : -1;
      pp += 64; col += dcol; hp += dslot;
      if (col >= PW) { col -= PW; ++hp; }
      while (hp >= PH) { hp -= PH; ++n; }
    }
  }
  auto issue_patch = [&](const int i, const int chunk, const int buf) {
    const unsigned voff = (poff[i] >= 0 && chunk < pa.nchunks) ? (unsigned)(poff[i] + chunk * 64) * 2u : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(smem + buf * PATCH + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, GIC_TRUNK_NT);
  };
  // ---- weight tile of one K step (tap, chunk): rows (tid >> 3) + 64 i of the BN output channels, 64 channels = 128 bytes each
  const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
  int b_off[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int n = bn0 + (tid >> 3) + i * 64;
    b_off[i] = n < N ? n * (int)d.ldb + kc : -1;
  }
  auto issue_b = [&](const int tap, const int chunk, const int st) {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const unsigned voff = (b_off[i] >= 0 && chunk < pa.nchunks) ? (unsigned)(b_off[i] + tap * Cin + chunk * 64) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(smem + RING0 + st * B_BYTES + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, 0);
    }
  };

  CSTAMP(1);
  // ---- prologue: patch of chunk 0, weight tiles of taps 0 and 1
#pragma unroll
  for (int i = 0; i < P; ++i) issue_patch(i, 0, 0);
  issue_b(0, 0, 0);
  issue_b(1, 0, 1);

  // fragment rows of this lane: output pixels bm0 + wr*32 + t*16 + frag_row(lr) -> patch pixel of tap (0, 0)
  int pbase[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = min(bm0 + wr * 32 + t * 16 + frag_row(lr), mlim - 1); // rows past the tile repeat its last pixel (never stored)
    const int n = fdiv(m, HW, iHW), rem = m - n * HW;
    const int ho = fdiv(rem, W, iW), wo = rem - ho * W;
    pbase[t] = (n * PH + ho - gp0) * PW + wo;
  }

  // ---- A-side BatchNorm + ReLU: [scale, shift] per input channel, then each thread normalises the pieces it DMA'd itself
  float* coef = (float*)(smem + COEF0);
  if constexpr (ABN) {
    for (int c = tid; c < Cin; c += NT) {
      const float gam = d.in_gamma[c], bet = d.in_beta[c];               // in flight together with the replicas
      float s1, s2;
      fold_replicas(d.in_stats, d.in_nrep, Cin, c, s1, s2);
      const float mean = s1 * d.in_inv_count;
      const float var = fmaxf(s2 * d.in_inv_count - mean * mean, 0.f);
      const float sc = gam * rsqrtf(var + 1e-5f);                        // kBnEps of encoder.hip (nn.BatchNorm2d default)
      coef[2 * c] = sc;
      coef[2 * c + 1] = bet - mean * sc;
    }
    __syncthreads();
  }
  auto normalise = [&](const int chunk, const int buf) {
    if constexpr (ABN) {
      // a thread's pieces all hold the same 8 channels of the chunk: q & 7 and (q >> 4) & 7 do not change with q += 512
      const int c16 = (tid & 7) ^ ((tid >> 4) & 7);
      const float4* cp = (const float4*)(coef + 2 * (chunk * 64 + c16 * 8));
      const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
      const float scl[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
      const float sft[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
      bf16x8 v[P];
#pragma unroll
      for (int i = 0; i < P; ++i) v[i] = *(const bf16x8*)(smem + buf * PATCH + (tid + NT * i) * 16);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        if (poff[i] < 0) continue;                                       // halo and unused pieces stay zero
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = (bf16_t)fmaxf((float)v[i][e] * scl[e] + sft[e], 0.f);
        *(bf16x8*)(smem + buf * PATCH + (tid + NT * i) * 16) = v[i];
      }
    }
  };
  CSTAMP(2);
  if constexpr (ABN) {
    wait_vm<2 * CB>();                                                   // the patch has landed (the two weight tiles may be in flight)
    CSTAMP(3);
    normalise(0, 0);
  }
  CSTAMP(4);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads / MFMAs of one 32-channel half of a K step
  auto read_half = [&](const unsigned char* patch, const int tapoff, const int st, const int ks, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
    const unsigned char* sB = smem + RING0 + st * B_BYTES;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int pp = pbase[t] + tapoff;
      fa[t] = *(const bf16x8*)(patch + pp * 128 + (((ks * 4 + lg) ^ ((pp >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int row = wc * (BN / 2) + t * 16 + lr;
      fb[t] = *(const bf16x8*)(sB + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
    }
  };
  auto mfma_half = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };

  // ---- K loop: chunk (64 channels) x tap (9, unrolled: ring stage = tap % 3 and every wait count is a constant).  Per step:
  //     wait (weight tile of this step) | barrier | read half 0 | mfma half 1 of the previous step | DMA: one patch piece of the
  //     next chunk (taps 0..P-1) + the weight tile two steps ahead | read half 1 | mfma half 0
  bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  for (int chunk = 0; chunk < pa.nchunks; ++chunk) {
    const int buf = MULTI ? (chunk & 1) : 0;
    const unsigned char* patch = smem + buf * PATCH;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // DMAs younger than this step's weight tile: the previous step's patch piece (if it issued one) and weight tile
      if (MULTI && ((t + 8) % 9) < P) wait_vm_lgkm<CB + 1>(); else wait_vm_lgkm<CB>();
      if (MULTI && t == 8 && chunk + 1 < pa.nchunks) normalise(chunk + 1, buf ^ 1);      // its pieces are older than this step's tile
      __builtin_amdgcn_s_barrier();
      const int tapoff = (t / 3) * PW + (t % 3);
      read_half(patch, tapoff, t % 3, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (chunk > 0 || t > 0) mfma_half(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (MULTI && t < P) issue_patch(t, chunk + 1, buf ^ 1);            // past the last chunk: zero fill into the idle buffer
      if (t + 2 < 9) issue_b(t + 2, chunk, (t + 2) % 3);
      else issue_b(t + 2 - 9, chunk + 1, (t + 2) % 3);                   // past the last chunk: zero fill, never read
      __builtin_amdgcn_sched_barrier(0);
      read_half(patch, tapoff, t % 3, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(fa0, fb0);
      // keep these MFMAs ABOVE the next step's wait + barrier: they cover the latency of the half-1 reads just issued (neither the
      // inline-asm wait nor s_barrier orders register-only instructions, and the scheduler sank seven of the eight below the barrier)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  mfma_half(fa1, fb1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // stray zero-fill DMAs must not land in the C tile
  __syncthreads();
  CSTAMP(5);

  // ---- epilogue (as tile8): C tile through LDS (16-byte row stores), BatchNorm column sums folded across the workgroup
  bf16_t* __restrict__ C = (bf16_t*)d.C;
  unsigned char* sC = smem;
  float* sStat = (float*)(smem + BM * SC);                               // [8 waves][BN/2][2]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wc * (BN / 2) + j * 16 + lr;
    float st_s = 0.f, st_q = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ml = wr * 32 + i * 16 + frag_row(lg * 4 + r);
        const float v = acc[i][j][r];
        *(bf16_t*)(sC + ml * SC + nl * 2) = (bf16_t)v;
        if (bm0 + ml < mlim) { st_s += v; st_q += v * v; }
      }
    }
    st_s += __shfl_xor(st_s, 16, 64); st_q += __shfl_xor(st_q, 16, 64);
    st_s += __shfl_xor(st_s, 32, 64); st_q += __shfl_xor(st_q, 32, 64);
    if (lg == 0) { sStat[(w * (BN / 2) + j * 16 + lr) * 2] = st_s; sStat[(w * (BN / 2) + j * 16 + lr) * 2 + 1] = st_q; }
  }
  __syncthreads();
  if (tid < BN) {                                                        // column tid: waves (wr = 0..3, wc)
    const int cwc = tid / (BN / 2), cl = tid % (BN / 2), n = bn0 + tid;
    if (n < N) {
      float s0 = 0.f, q0 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0 += sStat[((r * 2 + cwc) * (BN / 2) + cl) * 2];
        q0 += sStat[((r * 2 + cwc) * (BN / 2) + cl) * 2 + 1];
      }
      float* stp = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * N;
      atomicAdd(&stp[n], s0);
      atomicAdd(&stp[N + n], q0);
    }
  }
  constexpr int CPR = BN / 8;                                            // 16-byte chunks per tile row
  for (int c = tid; c < BM * CPR; c += NT) {
    const int ml = c / CPR, cc = c % CPR;
    const int m = bm0 + ml, n = bn0 + cc * 8;
    if (m < mlim && n < N) *(u32x4*)(C + (long)m * d.ldc + n) = *(const u32x4*)(sC + ml * SC + cc * 16);
  }
  CSTAMP(6);
}

// largest patch (pixels) over the row tiles of an [Nimg, H, W] output: global tiling (128 consecutive rows) or per-image tiling
int max_patch_pixels(int Nimg, int H, int W, bool per_image) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, bool>, int> cache;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(Nimg, H, W, per_image);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const long HW = (long)H * W, M = HW * Nimg;
  const int PH = H + 2, PW = W + 2;
  int best = 0;
  auto tile = [&](long bm0, long mlim) {
    const long n0 = bm0 / HW, h0 = (bm0 - n0 * HW) / W;
    const long n1 = (mlim - 1) / HW, h1 = (mlim - 1 - n1 * HW) / W;
    const int px = (int)((n1 * PH + h1 + 2 - (n0 * PH + h0) + 1) * PW);
    if (px > best) best = px;
  };
  if (per_image) {
    for (long b = 0; b < HW; b += 128) tile(b, b + 128 < HW ? b + 128 : HW);      // every image tiles alike
  } else {
    for (long b = 0; b < M; b += 128) tile(b, b + 128 < M ? b + 128 : M);
  }
  cache[key] = best;
  return best;
}

template <int BN, int P, bool MULTI, bool ABN>
bool launch_patch(const PatchDesc& pd, long tiles, hipStream_t stream) {
  constexpr size_t core = (size_t)(MULTI ? 2 : 1) * P * 8192 + 3 * BN * 128;
  const size_t lds = core + (ABN ? (size_t)pd.Cin * 8 : 0);
  static LdsGrant granted;
  if (!grant_lds(conv3x3_patch_kernel<BN, P, MULTI, ABN>, lds, granted)) return false;
  hipLaunchKernelGGL((conv3x3_patch_kernel<BN, P, MULTI, ABN>), dim3((unsigned)tiles), dim3(512), lds, stream, pd);
  return true;
}

template <int BN, bool ABN>
bool pick_patch(const PatchDesc& pd, int P, long tiles, hipStream_t stream) {
  if (pd.nchunks == 1) return P <= 6 ? launch_patch<BN, 6, false, ABN>(pd, tiles, stream) : (P <= 8 ? launch_patch<BN, 8, false, ABN>(pd, tiles, stream) : false);
  if (P <= 4) return launch_patch<BN, 4, true, ABN>(pd, tiles, stream);
  if (P == 5) return launch_patch<BN, 5, true, ABN>(pd, tiles, stream);
  if (P == 6) return launch_patch<BN, 6, true, ABN>(pd, tiles, stream);
  return false;
}

}  // namespace

bool try_conv3x3_patch(const GemmDesc& d, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_CONV3X3_PATCH") != nullptr;
  if (off || !d.conv || d.epi != EPI_BNSTATS || !d.stats) return false;
  if (d.in_dtype != DT_BF16 || d.out_dtype != DT_BF16) return false;
  if (d.cKH != 3 || d.cKW != 3 || d.cStride != 1 || d.cPad != 1 || d.cHo != d.cH || d.cWo != d.cW) return false;
  if (d.cCin % 64 || d.cCin > 1024 || d.N % 8 || d.ldc % 8 || (((uintptr_t)d.C) & 15) || (((uintptr_t)d.A) & 15) || (((uintptr_t)d.B) & 15)) return false;
  if (d.K != 9 * d.cCin || d.ldb % 8 || d.bias || d.alpha != 1.f || d.accumulate || d.res) return false;
  const bool abn = d.in_stats != nullptr;
  if (abn && (!d.in_gamma || !d.in_beta || d.in_inv_count <= 0.f || d.in_nrep < 1)) return false;
  const long HW = (long)d.cH * d.cW;
  if (HW <= 0 || d.M % HW || d.M >= (1 << 23)) return false;             // (the kernel's float-reciprocal divisions are exact below 2^23)
  const int Nimg = (int)(d.M / HW);
  const long a_elems = (long)d.M * d.cCin, b_elems = (long)(d.N - 1) * d.ldb + d.K;
  if (a_elems * 2 >= (1l << 31) || b_elems * 2 >= (1l << 31)) return false;
  PatchDesc pa;
  pa.A = d.A; pa.B = d.B; pa.C = d.C; pa.stats = d.stats;
  pa.in_stats = d.in_stats; pa.in_gamma = d.in_gamma; pa.in_beta = d.in_beta;
  pa.M = d.M; pa.N = d.N; pa.H = d.cH; pa.W = d.cW; pa.Cin = d.cCin; pa.ldb = (int)d.ldb; pa.ldc = (int)d.ldc;
  pa.stats_nrep = d.stats_nrep < 1 ? 1 : d.stats_nrep; pa.in_nrep = d.in_nrep; pa.in_inv_count = d.in_inv_count;
  pa.nchunks = d.cCin / 64;
  // tiles that never cross an image need two halo rows less; taken where the short last tile of an image costs <= 5 % more tiles
  const int tpi = cdiv(HW, 128);
  const bool per_image = (double)tpi * 128 <= 1.05 * (double)HW;
  const int max_pix = max_patch_pixels(Nimg, d.cH, d.cW, per_image);
  const int P = (max_pix + 63) / 64;
  pa.tpi = per_image ? tpi : 0;
  pa.tiles_m = per_image ? tpi * Nimg : cdiv(d.M, 128);
  // output-channel tile: 128 where that still leaves about a workgroup per CU, else 64
  const bool n128 = d.N >= 128 && (long)pa.tiles_m * cdiv(d.N, 128) >= 160;
  const long tiles = (long)pa.tiles_m * (n128 ? cdiv(d.N, 128) : cdiv(d.N, 64));
  pa.a_bytes = (unsigned)(a_elems * 2); pa.b_bytes = (unsigned)(b_elems * 2);
  const int tiles_n = n128 ? cdiv(d.N, 128) : cdiv(d.N, 64);
  pa.tiles_n = xcd_share_a(a_elems * 2, b_elems * 2, tiles_n) ? tiles_n : 0;
  if (abn) return n128 ? pick_patch<128, true>(pa, P, tiles, stream) : pick_patch<64, true>(pa, P, tiles, stream);
  return n128 ? pick_patch<128, false>(pa, P, tiles, stream) : pick_patch<64, false>(pa, P, tiles, stream);
}

}  // namespace gic
#ifdef GIC_STAMPS
extern "C" int gic_debug_conv3x3_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gic::g_cstamp), sizeof(unsigned long long) * 2 * 8 * 2);
}
#endif
