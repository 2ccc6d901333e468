// The stem of the ResNet trunk (torchvision conv1: 7x7 / stride 2 / pad 3, 3 -> 64 channels, reference src/generator.py:12-14) on the
// zero-bordered NHWC4 image that gic_pack_image writes, as a STREAMING kernel: window [7 x 8 x 4] (the eighth tap column and the fourth
// channel are zero), so one 32-deep MFMA k-step = one window row, and a lane's 8 k-values = two adjacent pixels = 16 contiguous bytes.
//
// tile8 gathers this layer's A operand 16 bytes at a time from L2 (every output pixel's 7 x 64-byte window: 360 MB of gathers for
// a 27 MB image batch) and ran it at 1.7 TB/s of algorithmic traffic: 76 us per launch against an HBM floor of 16 us (130 MB, four
// fifths of it the output).  Here a persistent workgroup owns a range of output rows of one image: the input rows it needs roll
// through an LDS ring by LDS-DMA (two new rows per output row, three output rows ahead), the 28 KB of weights stay in LDS, every
// window is read from the ring, the BatchNorm column sums stay in registers across the workgroup's rows (one atomic per column at the
// end), and the C tile is double-buffered so that a row's stores run under the next row's MFMAs (as conv1x1_stream.hip).
#include <stdlib.h>

#include "conv_stem.h"

namespace gic {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct StemDesc {
  const void* A; const void* B; void* C; float* stats;
  int Nimg, H, W, Ho, Wo, ldc, stats_nrep;
  int ranges, rows_per_range;            // workgroups per image, output rows per workgroup
  unsigned a_bytes, b_bytes;
};

constexpr int kRing = 16;                // input-row slots (7 in use + 8 in flight)
constexpr int kSlot = 2048;              // bytes per slot: two whole LDS-DMA wave instructions (a row is W * 8 <= 2048 bytes)
constexpr int kAhead = 3;                // output rows whose input rows are in flight beyond the prologue's

__global__ __launch_bounds__(512) void conv_stem_kernel(const StemDesc d) {
  constexpr int NT = 512, BN = 64, KR = 7;
  constexpr int W_BYTES = KR * 4 * BN * 16;                             // [r][k-group][n][16 B] = 28 KB
  constexpr int SC = BN * 2 + 16, C_BYTES = 128 * SC;
  constexpr int W0 = kRing * kSlot, C0 = W0 + W_BYTES, ST0 = C0 + 2 * C_BYTES;
  constexpr int CS = 2;                                                 // 16-byte stores of a C tile per thread (Wo * 8 <= 1024 pieces)
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w & 3, wm = w >> 2;                                    // wave -> 16 output channels, one half of the row's pixel tiles
  const int lr = lane & 15, lg = lane >> 4;
  const int n_img = blockIdx.x / d.ranges, rg = blockIdx.x % d.ranges;
  const int h0 = rg * d.rows_per_range, h1 = min(h0 + d.rows_per_range, d.Ho);
  const int W = d.W, Wo = d.Wo;
  const int row_pieces = W * 8 / 16;                                    // 16-byte pieces of an input row (W even)

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)d.a_bytes, 0x00020000);
  // input row j (of this image) -> ring slot j % kRing; issued by waves 0..3 only: waves 2i, 2i+1 bring row `first + i`
  auto issue_rows = [&](const int first) {
    if (w < 4) {
      const int j = first + (w >> 1);
      const int piece = (w & 1) * 64 + lane;
      const unsigned voff = (j < d.H && piece < row_pieces) ? (unsigned)((n_img * d.H + j) * W * 4 + piece * 8) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(smem + (j % kRing) * kSlot + (w & 1) * 1024), 16, (int)voff, 0, 0, GIC_TRUNK_NT);
    }
  };
  // ---- prologue: the rows of the first 1 + kAhead output rows (7 + 2 kAhead = 13, in pairs), the weights
  for (int j = 2 * h0; j < 2 * h0 + 7 + 2 * kAhead; j += 2) issue_rows(j);
  {
    // weights [64][7][8][4] bf16 (k = r*32 + s*4 + c) -> LDS [r][k-group lg][n][16 B]: a B fragment read is 16 lanes x 16 contiguous bytes
    const bf16_t* Wg = (const bf16_t*)d.B;
    for (int i = tid; i < KR * 4 * BN; i += NT) {
      const int n = i % BN, lgk = (i / BN) % 4, r = i / (4 * BN);
      *(u32x4*)(smem + W0 + i * 16) = *(const u32x4*)(Wg + (long)n * (KR * 32) + r * 32 + lgk * 8);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float st_s[4], st_q[4];                                               // BatchNorm sums of this lane's column, per pixel tile slot
#pragma unroll
  for (int i = 0; i < 4; ++i) st_s[i] = st_q[i] = 0.f;
  bf16_t* __restrict__ C = (bf16_t*)d.C;
  unsigned char* sC = smem + C0;
  auto store_row = [&](const int ho, const unsigned char* buf) {
    bf16_t* dst = C + ((long)(n_img * d.Ho + ho) * Wo) * d.ldc;
#pragma unroll
    for (int i = 0; i < CS; ++i) {
      const int c = tid + NT * i;
      const int ml = c >> 3, cc = c & 7;
      const u32x4 v = *(const u32x4*)(buf + ml * SC + cc * 16);
      if (ml < Wo) *(u32x4*)(dst + (long)ml * d.ldc + cc * 8) = v;
    }
  };

  int jj = 0;
  for (int ho = h0; ho < h1; ++ho, ++jj) {
    // The rows of output row `ho` were issued kAhead + 1 iterations ago (or by the prologue, which drained): behind them came that
    // iteration's 2 stores and kAhead iterations' DMA + 2 stores (waves 0..3; the others issue stores only and wait at the barrier).
    wait_vm<2 + 3 * kAhead>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // rows published; the previous C tile is complete; every wave is done with the previous row's ring slots
    issue_rows(2 * (ho + kAhead + 1) + 5);                               // the two new rows of output row ho + kAhead + 1
    unsigned char* sCj = sC + (jj & 1) * C_BYTES;
    if (jj > 0) store_row(ho - 1, sC + ((jj - 1) & 1) * C_BYTES);        // runs under this row's MFMAs
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const unsigned char* row = smem + ((2 * ho + r) % kRing) * kSlot;
      const bf16x8 fb = *(const bf16x8*)(smem + W0 + ((r * 4 + lg) * BN + wn * 16 + lr) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int wo = (wm * 4 + i) * 16 + lr;                           // pixel tiles past Wo read the slot's tail (discarded)
        const bf16x8 fa = *(const bf16x8*)(row + ((2 * wo + 2 * lg) & 255) * 8);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
      }
    }
    // accumulator row 4 lg + q = pixel, column lr = channel wn*16 + lr
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ml = (wm * 4 + i) * 16 + lg * 4 + q;
        const float v = acc[i][q];
        *(bf16_t*)(sCj + ml * SC + (wn * 16 + lr) * 2) = (bf16_t)v;
        if (ml < Wo) { st_s[i] += v; st_q[i] += v * v; }
      }
    }
  }
  __syncthreads();
  if (jj > 0) store_row(h1 - 1, sC + ((jj - 1) & 1) * C_BYTES);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- BatchNorm column sums: fold the lane's four pixel-tile slots, the 4 row groups of the wave, the two pixel halves; one atomic per column
  float s = st_s[0] + st_s[1] + st_s[2] + st_s[3], q = st_q[0] + st_q[1] + st_q[2] + st_q[3];
  s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
  s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
  float* sStat = (float*)(smem + ST0);                                   // [2 pixel halves][64][2]
  if (lg == 0) { sStat[(wm * BN + wn * 16 + lr) * 2] = s; sStat[(wm * BN + wn * 16 + lr) * 2 + 1] = q; }
  __syncthreads();
  if (tid < BN) {
    float* stp = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * BN;
    atomicAdd(&stp[tid], sStat[tid * 2] + sStat[(BN + tid) * 2]);
    atomicAdd(&stp[BN + tid], sStat[tid * 2 + 1] + sStat[(BN + tid) * 2 + 1]);
  }
}

}  // namespace

bool try_conv_stem(const GemmDesc& d, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_CONV_STEM") != nullptr;
  if (off || !d.conv || d.epi != EPI_BNSTATS || !d.stats || d.in_stats || d.res) return false;
  if (d.in_dtype != DT_BF16 || d.out_dtype != DT_BF16) return false;
  if (d.cKH != 7 || d.cKW != 8 || d.cCin != 4 || d.cStride != 2 || d.cPad != 0 || d.N != 64 || d.K != 224 || d.ldb != 224) return false;
  if (d.cW % 2 || d.cW * 8 > kSlot || d.cWo > 128 || d.cWo < 1 || d.ldc % 8) return false;
  if (2 * (d.cWo - 1) + 8 > d.cW || 2 * (d.cHo - 1) + 7 > d.cH) return false;                 // windows inside the (pre-padded) image
  if ((((uintptr_t)d.C) & 15) || (((uintptr_t)d.A) & 15) || (((uintptr_t)d.B) & 15) || d.bias || d.alpha != 1.f || d.accumulate) return false;
  const long HoWo = (long)d.cHo * d.cWo;
  if (HoWo <= 0 || d.M % HoWo) return false;
  StemDesc sd;
  sd.Nimg = (int)(d.M / HoWo);
  const long a_elems = (long)sd.Nimg * d.cH * d.cW * 4, b_elems = 64l * 224;
  if (a_elems * 2 >= (1l << 31)) return false;
  // about one persistent workgroup per CU: split every image's output rows into ranges
  int ranges = 256 / sd.Nimg;
  if (ranges < 1) ranges = 1;
  if (ranges > d.cHo) ranges = d.cHo;
  sd.rows_per_range = cdiv(d.cHo, ranges);
  sd.ranges = cdiv(d.cHo, sd.rows_per_range);
  sd.A = d.A; sd.B = d.B; sd.C = d.C; sd.stats = d.stats;
  sd.H = d.cH; sd.W = d.cW; sd.Ho = d.cHo; sd.Wo = d.cWo; sd.ldc = (int)d.ldc;
  sd.stats_nrep = d.stats_nrep < 1 ? 1 : d.stats_nrep;
  sd.a_bytes = (unsigned)(a_elems * 2); sd.b_bytes = (unsigned)(b_elems * 2);
  constexpr size_t lds = (size_t)kRing * kSlot + 7 * 4 * 64 * 16 + 2 * 128 * (64 * 2 + 16) + 2 * 64 * 2 * 4;
  static LdsGrant granted;
  if (!grant_lds(conv_stem_kernel, lds, granted)) return false;
  hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)(sd.Nimg * sd.ranges)), dim3(512), lds, stream, sd);
  return true;
}

}  // namespace gic
