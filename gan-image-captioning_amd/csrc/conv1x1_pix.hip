// conv3 of the 14x14 and 7x7 bottlenecks of the ResNet trunk (1x1, K = 256 input channels into 1024 / K = 512 into 2048, BatchNorm +
// ReLU of the input on load; reference src/generator.py:12-14) with the PIXELS resident in registers (round 3).
//
// The panel kernel (conv1x1_panel.hip) keeps the A panel of a 128-row tile in LDS and runs its output-channel tiles through a
// two-stage weight ring; per 64-channel tile it pays fragment reads of BOTH operands, a C tile staged through LDS (2-byte writes),
// a second barrier and the row stores: 21-25 us per launch against an HBM floor of 4.  Here the products are transposed as in
// conv_b2b.hip: a wave owns 16 pixels, their K channels are the MFMA B operand and stay in registers (K / 32 fragments per lane,
// loaded from global memory once and normalised there), the weights are the A operand and are the only thing in LDS (a three-stage
// LDS-DMA ring of 64-channel tiles, rows laid out so that a lane ends up with 16 CONSECUTIVE output channels of its pixel): 32-byte
// stores straight from the accumulators, one barrier per tile, no C tile.  The BatchNorm column sums of a tile: the wave's 64 x 16
// tile is turned through a private LDS scratch (lane c then holds channel c of all 16 pixels), summed, kept in registers per tile and
// folded across the eight waves once at the end (what the butterfly and the LDS atomics of the first versions cost).
// K = 512: two 64 KB stages leave no room for the scratch; its column sums take the reduce-scatter butterfly of conv_b2b.hip on DPP.
#include <stdlib.h>

#include "conv1x1_pix.h"
#include "bn_fold.h"

namespace gic {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 pix_load16(const __amdgpu_buffer_rsrc_t r, const int voff, const int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ void pix_store16(const u32x4 v, const __amdgpu_buffer_rsrc_t r, const int voff, const int soff) {
  // The tile offset rides in the per-lane offset, not in the scalar one: the compiler (hipcc 7.2) assumes a store of more than 8 bytes
  // with an SGPR offset needs no wait state before a VALU instruction overwrites its data registers and schedules one right behind
  // it; on gfx950 that instruction's result reached memory in place of the first dword (sporadically, lanes 12-15 of each row of 16).
#ifdef GIC_STORE_SOFF                                                      // (measurement build: the form that exposes the hazard)
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
#else
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, 0);
#endif
}
// lane l's value of its row-of-16 neighbour l ^ X on the VALU (DPP), and the reduce-scatter butterfly of conv_b2b.hip on them: the
// column sums where the turning scratch does not fit the LDS (K = 512)
template <int CTRL>
__device__ __forceinline__ float pix_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float pix_xor4(const float v) {                 // lanes with bit 2 clear read l + 4 (row_ror:12, banks 0 and 2), the others l - 4
  const int x = __builtin_bit_cast(int, v);
  int a = __builtin_amdgcn_update_dpp(0, x, 0x12C, 0xF, 0x5, false);
  a = __builtin_amdgcn_update_dpp(a, x, 0x124, 0xF, 0xA, false);
  return __builtin_bit_cast(float, a);
}
__device__ __forceinline__ float pix_reduce_scatter16(const float (&v)[16], const int lr) {   // sum over the row's 16 lanes, value e landing in lane lr == e
  float t[8], u[4], x[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const bool up = lr & 8; t[i] = (up ? v[i + 8] : v[i]) + pix_dpp<0x128>(up ? v[i] : v[i + 8]); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { const bool up = lr & 4; u[i] = (up ? t[i + 4] : t[i]) + pix_xor4(up ? t[i] : t[i + 4]); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const bool up = lr & 2; x[i] = (up ? u[i + 2] : u[i]) + pix_dpp<0x4E>(up ? u[i] : u[i + 2]); }
  const bool up = lr & 1;
  return (up ? x[1] : x[0]) + pix_dpp<0xB1>(up ? x[0] : x[1]);
}

struct PixDesc {
  const void* A; const void* B; void* C; float* stats;
  const float* in_stats; const float* in_gamma; const float* in_beta;
  int M, N, stats_nrep, in_nrep;
  float in_inv_count;
  int tiles_m, per_group;              // row tiles of 128 pixels; 64-channel tiles per workgroup (grid = tiles_m * groups)
  unsigned a_bytes, b_bytes, c_bytes;
};

// K input channels; NSTG ring stages of 64-channel weight tiles; TURN: column sums through the turning scratch (else: DPP butterfly)
template <int K, int NSTG, bool TURN>
__global__ __launch_bounds__(512) void conv1x1_pix_kernel(const PixDesc d) {
  constexpr int NT = 512, KS = K / 32;
  constexpr int ROWB = K * 2, CH = ROWB / 16;                            // bytes / 16-byte pieces of a weight row
  constexpr int W_BYTES = 64 * ROWB, PW = W_BYTES / 16 / NT;             // a 64-channel tile: 32 | 64 KB, 4 | 8 pieces per thread
  constexpr int COEF0 = NSTG * W_BYTES;
  constexpr int MAXT = 8;                                                // 64-channel tiles per workgroup at most (the host's per_group)
  static_assert(PW <= 8, "piece offset array");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* coef = (float*)(smem + COEF0);                                  // [K][2] scale, shift of the input's BatchNorm
  float* sT = coef + 2 * K;                                  // [8 waves][16 pixels][68] f32: a wave's tile, turned (below)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int tile_m = blockIdx.x % d.tiles_m, grp = blockIdx.x / d.tiles_m;
  const int nt0 = grp * d.per_group, ntiles_n = d.N / 64;
  const int nt1 = min(nt0 + d.per_group, ntiles_n);
  const int ntl = nt1 - nt0;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)d.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, (int)d.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(d.C, 0, (int)d.c_bytes, 0x00020000);
  const int wbase = (tid & ~63) * 16;
  // weight tile image [64 rows][K]: LDS row L = 16 j + x holds output channel 16 (x >> 2) + 4 j + (x & 3) of the tile; 16-byte pieces
  // XOR-swizzled by L & 15 within each 256 bytes of a row (rows are 512 / 1024 bytes: all start on bank 0)
  int w_off[8];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int q = tid + NT * i, L = q / CH, slot = q % CH;
    const int j = L >> 4, x = L & 15, ch = 16 * (x >> 2) + 4 * j + (x & 3);
    const int kp = slot ^ (L & 15);
    w_off[i] = (ch * K + kp * 8) * 2;
  }
  auto issue_w_ = [&](const int t, const int st_, const int (&wo)[8]) {   // weight tile t of this workgroup's range (past it: zero fill, no traffic)
    const bool ok = t < ntl;
    const int so = ok ? (nt0 + t) * (64 * K * 2) : 0, oob = ok ? 0 : (int)0x80000000;
#pragma unroll
    for (int i = 0; i < PW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(smem + st_ * W_BYTES + i * (NT * 16) + wbase), 16, wo[i] | oob, so, 0, 0);
  };
  auto issue_w = [&](const int t, const int st_) { issue_w_(t, st_, w_off); };
#pragma unroll
  for (int s_ = 0; s_ < NSTG - 1; ++s_) issue_w(s_, s_);

  // ---- this lane's pixel: its K channels as B-operand fragments (k slots lg * 8 .. + 7 of each 32-deep slice)
  const int m = tile_m * 128 + w * 16 + lr;
  const bool mok = m < d.M;
  const int mc = mok ? m : d.M - 1;                                      // rows past M: a pixel of zeros after the normalisation (nothing summed), not stored
  bf16x8 fy[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fy[ks] = __builtin_bit_cast(bf16x8, pix_load16(rsA, (mc * K + lg * 8) * 2, ks * 64));

  // ---- the input's BatchNorm coefficients, the column-sum table
  for (int c = tid; c < K; c += NT) {
    float s1, s2;
    fold_replicas(d.in_stats, d.in_nrep, K, c, s1, s2);
    const float mean = s1 * d.in_inv_count, var = fmaxf(s2 * d.in_inv_count - mean * mean, 0.f);
    const float sc = d.in_gamma[c] * rsqrtf(var + 1e-5f);                // kBnEps of encoder.hip (nn.BatchNorm2d default)
    coef[2 * c] = sc; coef[2 * c + 1] = d.in_beta[c] - mean * sc;
  }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float4* cp = (const float4*)(coef + 2 * (ks * 32 + lg * 8));
    const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
    const float scl[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
    const float sft[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) fy[ks][e] = (bf16_t)fmaxf((float)fy[ks][e] * scl[e] + sft[e], 0.f);
    if (!mok) fy[ks] = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
  }

  // byte offset of this lane's 16 channels inside a tile of its pixel's row (rows past M: past the descriptor's extent, stores dropped)
  const int coff = mok ? (m * d.N + lg * 16) * 2 : (int)0x80000000;
  // Column sums: the wave's 64 x 16 tile goes through LDS once and comes back turned -- lane c with channel c of all 16 pixels -- so
  // a sum costs one add per value.  (In registers it is a 16-lane butterfly of selects and cross-lane adds per value: ~120 VALU
  // instructions per tile and wave against 32 MFMAs, 13 of the first version's 32 us.)  Rows of 68 floats: the 16-byte writes of
  // the 16 pixels fall on different banks, the reads are lane-contiguous.  Written and read by the same wave: no barrier.
  const unsigned st_w = (unsigned)(__SIZE_TYPE__)(lds_void_ptr)sT + w * (16 * 272) + lr * 272 + lg * 64;
  const unsigned st_r = (unsigned)(__SIZE_TYPE__)(lds_void_ptr)sT + w * (16 * 272) + lane * 4;
  float rs[MAXT], rq[MAXT];                                              // lane c: sums of channel c of tile i over this wave's 16 pixels
#pragma unroll
  for (int i = 0; i < MAXT; ++i) rs[i] = rq[i] = 0.f;
  int st = 0;
  for (int t = 0; t < ntl; ++t) {
    // Weight tile t (issued NSTG - 1 tiles ago) has landed once only what was issued behind it is outstanding: per tile in between its
    // PW pieces and two stores (one in-order counter for loads, LDS-DMA and stores); the first tiles have fewer stores behind them --
    // the stricter count is still correct there.
    if constexpr (NSTG == 3) {
      if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
      else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + 4) : "memory");
    } else {
      if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                        // tile t visible; the stage of tile t - 1 is free
    asm volatile("" ::: "memory");
    issue_w(t + NSTG - 1, st == 0 ? NSTG - 1 : st - 1);
    const unsigned char* sW = smem + st * W_BYTES;
    // block j, MFMA row x <-> output channel 16 (x >> 2) + 4 j + (x & 3) of the tile: lane (lr, lg) ends up with channels 16 lg + 4 j + r
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int L = 16 * j + lr, kp = ks * 4 + lg;
        const bf16x8 fw = *(const bf16x8*)(sW + L * ROWB + ((kp ^ lr) << 4));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fy[ks], acc[j], 0, 0, 0);
      }
    }
    // ---- 16 consecutive output channels 64 (nt0 + t) + 16 lg + e of this pixel: 32-byte store, column sums
    bf16x8 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e >> 3][e & 7] = (bf16_t)acc[e >> 2][e & 3];
    pix_store16(__builtin_bit_cast(u32x4, o[0]), rsC, coff, (nt0 + t) * 128);
    pix_store16(__builtin_bit_cast(u32x4, o[1]), rsC, coff + 16, (nt0 + t) * 128);
    float s, q;
    if constexpr (TURN) {
    // LDS traffic issued behind the compiler's back: it would make an LDS write it knows of wait for every LDS-DMA in flight
    // (may-alias), which is the ring this loop keeps ahead.  (Behind the stores: their conversions have read the accumulators.)
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\tds_write_b128 %0, %4 offset:48"
                 ::"v"(st_w), "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    float x[16];
    asm volatile("ds_read_b32 %0, %16\n\tds_read_b32 %1, %16 offset:272\n\tds_read_b32 %2, %16 offset:544\n\tds_read_b32 %3, %16 offset:816\n\t"
                 "ds_read_b32 %4, %16 offset:1088\n\tds_read_b32 %5, %16 offset:1360\n\tds_read_b32 %6, %16 offset:1632\n\tds_read_b32 %7, %16 offset:1904\n\t"
                 "ds_read_b32 %8, %16 offset:2176\n\tds_read_b32 %9, %16 offset:2448\n\tds_read_b32 %10, %16 offset:2720\n\tds_read_b32 %11, %16 offset:2992\n\t"
                 "ds_read_b32 %12, %16 offset:3264\n\tds_read_b32 %13, %16 offset:3536\n\tds_read_b32 %14, %16 offset:3808\n\tds_read_b32 %15, %16 offset:4080\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
                   "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15])
                 : "v"(st_r));
    s = 0.f; q = 0.f;
#pragma unroll
    for (int p_ = 0; p_ < 16; ++p_) { s += x[p_]; q += x[p_] * x[p_]; }
    } else {
      // lane (lr, lg) ends up with the sums of channel 16 lg + lr = its lane index: the same slot the turned form fills
      float vs[16], vq[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) { vs[e] = acc[e >> 2][e & 3]; vq[e] = vs[e] * vs[e]; }
      s = pix_reduce_scatter16(vs, lr); q = pix_reduce_scatter16(vq, lr);
    }
    // kept in registers until the end (tile t's pair in slot t): LDS float atomics cost 9 of the first version's 29 us
#pragma unroll
    for (int i = 0; i < MAXT; ++i) { rs[i] = i == t ? s : rs[i]; rq[i] = i == t ? q : rq[i]; }
    st = st == NSTG - 1 ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            // trailing zero-fill DMAs have landed: the ring is free
  __syncthreads();
  float2* red = (float2*)smem;                                           // [8 waves][MAXT][64] (sum, sum of squares): 32 KB of the ring
#pragma unroll
  for (int i = 0; i < MAXT; ++i) red[(w * MAXT + i) * 64 + lane] = make_float2(rs[i], rq[i]);
  __syncthreads();
  float* stp = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * d.N;
  for (int i = tid; i < ntl * 64; i += NT) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) { const float2 v = red[(ww * MAXT + (i >> 6)) * 64 + (i & 63)]; s1 += v.x; s2 += v.y; }
    atomicAdd(&stp[nt0 * 64 + i], s1);
    atomicAdd(&stp[d.N + nt0 * 64 + i], s2);
  }
}

template <int K, int NSTG, bool TURN>
bool launch_pix(const PixDesc& pd, int groups, hipStream_t stream) {
  const size_t lds = (size_t)NSTG * 64 * K * 2 + (size_t)K * 8 + (TURN ? 8 * 16 * 272 : 0);
  if (lds > 160 * 1024) return false;
  static LdsGrant granted;
  if (!grant_lds(conv1x1_pix_kernel<K, NSTG, TURN>, lds, granted)) return false;
  hipLaunchKernelGGL((conv1x1_pix_kernel<K, NSTG, TURN>), dim3((unsigned)(pd.tiles_m * groups)), dim3(512), lds, stream, pd);
  return true;
}

}  // namespace

bool try_conv1x1_pix(const GemmDesc& d, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_CONV1X1_PIX") != nullptr;
  if (off || !d.conv || d.epi != EPI_BNSTATS || !d.stats || d.res || !d.in_stats) return false;      // (the input's BatchNorm rides in: conv3 of a bottleneck)
  if (d.in_dtype != DT_BF16 || d.out_dtype != DT_BF16) return false;
  if (d.cKH != 1 || d.cKW != 1 || d.cStride != 1 || d.cPad != 0) return false;
  if ((d.K != 256 && d.K != 512) || d.cCin != d.K || d.lda != d.K || d.ldb != d.K || d.N < 512 || d.N % 64 || d.ldc != d.N || d.M < 128) return false;
  if ((((uintptr_t)d.C) & 15) || (((uintptr_t)d.A) & 15) || (((uintptr_t)d.B) & 15)) return false;
  if (d.bias || d.alpha != 1.f || d.accumulate || d.stats_only) return false;
  if (!d.in_gamma || !d.in_beta || d.in_inv_count <= 0.f || d.in_nrep < 1) return false;
  const long a_bytes = (long)d.M * d.K * 2, b_bytes = (long)d.N * d.K * 2, c_bytes = (long)d.M * d.N * 2;
  if (a_bytes >= (1l << 31) || b_bytes >= (1l << 31) || c_bytes >= (1l << 31)) return false;
  PixDesc pd;
  pd.tiles_m = cdiv(d.M, 128);
  const int tiles_n = d.N / 64;
  // about one workgroup per CU (a workgroup keeps its pixels in registers: every further group of a row tile loads them again)
  static const int wg_target = [] { const char* e = getenv("GIC_PIX_WG"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
  int groups = wg_target / pd.tiles_m;
  if (groups < 1) groups = 1;
  if (groups > tiles_n) groups = tiles_n;
  if (cdiv(tiles_n, groups) > 8) groups = cdiv(tiles_n, 8);            // (the kernel's MAXT)
  pd.per_group = cdiv(tiles_n, groups);
  groups = cdiv(tiles_n, pd.per_group);
  pd.A = d.A; pd.B = d.B; pd.C = d.C; pd.stats = d.stats;
  pd.in_stats = d.in_stats; pd.in_gamma = d.in_gamma; pd.in_beta = d.in_beta;
  pd.M = d.M; pd.N = d.N; pd.stats_nrep = d.stats_nrep < 1 ? 1 : d.stats_nrep; pd.in_nrep = d.in_nrep; pd.in_inv_count = d.in_inv_count;
  pd.a_bytes = (unsigned)a_bytes; pd.b_bytes = (unsigned)b_bytes; pd.c_bytes = (unsigned)c_bytes;
  // (K = 512: two 64 KB stages leave no room for the turning scratch: its column sums take the butterfly)
  return d.K == 256 ? launch_pix<256, 3, true>(pd, groups, stream) : launch_pix<512, 2, false>(pd, groups, stream);
}

}  // namespace gic
