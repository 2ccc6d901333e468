// MFMA GEMM family for gfx950 (MI355X): every dense contraction on the hot path (trunk convolutions as implicit GEMM, LSTM
// gates, vocab projection, discriminator embedding / highway / head, all dgrad and wgrad products, the encoder head).
//
// Two kernels:
//   gemm_kernel   4 waves (2x2), 128x128 or 64x64 tile.  Any operand layout (k-contiguous: LDS image [row][k], ds_read_b128
//                 fragments; m/n-contiguous: LDS image [k][row], bf16 fragments out of ds_read_b64_tr_b16, so dgrad / wgrad
//                 need no transposed copies in HBM), bf16 (v_mfma_f32_16x16x32_bf16) or exact f32 (v_mfma_f32_16x16x4_f32,
//                 parity mode), register-staged double buffer or 3-stage LDS-DMA ring, split-K over gridDim.y (f32 atomics)
//                 for skinny / deep-K products, highway epilogue, 16-byte accesses when shapes allow, scalar fallback.
//   tile8_kernel  8 waves (4x2), 128x128 or 128x64 tile, bf16 k-contiguous operands only: buffer-descriptor LDS-DMA ring of 1 /
//                 2 / 4 stages.  Every trunk convolution in bf16 mode and the wide plain / highway products.  See its header.
// Both: XCD-aware block -> tile map (each of the 8 XCDs owns a contiguous run of tiles that share B panels in its L2), C tile
// staged through LDS for 16-byte row stores, BatchNorm column sums folded across the block in the epilogue.
#include "gemm.h"
#include "conv3x3.h"
#include "conv1x1_stream.h"
#include "conv1x1_panel.h"
#include "conv1x1_pix.h"
#include "conv_stem.h"
#include "bn_fold.h"

#include <stdlib.h>

namespace gic {

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;

// 16 zero bytes in global memory: the source of an LDS-DMA chunk that falls outside the operand (conv padding, M/K tails)
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

#ifdef GIC_STAMPS
}  // anon
__device__ unsigned long long g_stamp[32];
namespace {
#define STAMP(i) do { if (tid == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) { \
  const int o_ = blockIdx.x == 0 ? 0 : 16; g_stamp[o_ + 2 * (i)] = __builtin_amdgcn_s_memtime(); g_stamp[o_ + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define DBG d.dbg      // phase ablation (tools/gemm_stamps.py): 1 = no DMA after the prologue, 2 = no MFMA phase
#else
#define STAMP(i) do {} while (0)
#define DBG 0
#endif

template <typename T> struct GlobalPtr { typedef const __attribute__((address_space(1))) T* type; };

// PIPE (k-contiguous, vectorised operands only): tiles reach LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPR
// staging) into a 3-stage ring; the loads of tile k+2 are in flight under the MFMAs of tiles k and k+1 behind a COUNTED
// s_waitcnt vmcnt and one raw s_barrier per K tile.  The LDS image is lane-linear per wave-instruction (128-byte rows, no
// padding), so the bank-conflict fix is an XOR of the 16-byte chunk index with (row>>1)&7, applied to the per-lane SOURCE
// address on the way in and to the ds_read_b128 address on the way out.
template <typename TI, typename TO, bool AKC, bool BKC, int BM, int BN, bool VEC, int EPI, bool CONV, bool PIPE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmDesc d, const int k_tiles_per_split) {
  static_assert(!CONV || (AKC && BKC && VEC), "the implicit-GEMM convolution loader is k-contiguous and vectorised");
  static_assert(!PIPE || (AKC && BKC && VEC), "the LDS-DMA pipeline needs k-contiguous 16-byte chunks");
  constexpr int SZ = sizeof(TI);
  constexpr int BK = 128 / SZ;          // K elements per tile
  constexpr int VE = 16 / SZ;           // elements per 16-byte chunk
  constexpr int SA = PIPE ? 128 : (AKC ? 144 : BM * SZ + 16);   // LDS row stride (bytes)
  constexpr int SB = PIPE ? 128 : (BKC ? 144 : BN * SZ + 16);
  constexpr int NSTAGE = PIPE ? 3 : 2;
  constexpr int A_BYTES = AKC ? BM * SA : BK * SA;
  constexpr int B_BYTES = BKC ? BN * SB : BK * SB;
  constexpr int BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int TM = BM / 32, TN = BN / 32;      // 16x16 tiles per wave
  constexpr int CA = BM / 32, CB = BN / 32;      // 16-B chunks per thread per tile
  constexpr bool IS_BF16 = (SZ == 2);
  typedef typename GlobalPtr<TI>::type gptr_t;

  constexpr int EPI_BYTES = BM * (BN * (int)sizeof(TO) + 16) + 4 * (BN / 2) * 2 * 4;   // staged C tile + stats scratch
  constexpr int SMEM_BYTES = NSTAGE * BUF_BYTES > EPI_BYTES ? NSTAGE * BUF_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  STAMP(0);
  const int wr = w >> 1, wc = w & 1;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- XCD-aware tile assignment (bijective remap; blocks b and b+8 share an XCD)
  const int tiles_m = (d.M + BM - 1) / BM;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int bm0 = (bid % tiles_m) * BM;
  const int bn0 = (bid / tiles_m) * BN;

  gptr_t A = (gptr_t)d.A;
  gptr_t B = (gptr_t)d.B;
  const int M = d.M, N = d.N, K = d.K;
  const long lda = d.lda, ldb = d.ldb;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[CA], rb[CB];

  // ---- implicit-GEMM convolution: per staged row (fixed per thread) the image and the top-left input pixel;
  // per thread ONE running (r, s, c) decomposition of its k (all of a thread's chunks share kc = tid & 7).
  long cv_base[CONV ? CA : 1];
  int cv_hi0[CONV ? CA : 1], cv_wi0[CONV ? CA : 1];
  int cv_r = 0, cv_s = 0, cv_c = 0;
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int m = bm0 + ((tid + i * 256) >> 3);
      const int mm = m < M ? m : 0;
      const int wo = mm % d.cWo, t = mm / d.cWo;
      const int ho = t % d.cHo, n = t / d.cHo;
      cv_hi0[i] = m < M ? ho * d.cStride - d.cPad : -(1 << 28);      // rows past M never validate
      cv_wi0[i] = wo * d.cStride - d.cPad;
      cv_base[i] = ((long)n * d.cH + cv_hi0[i]) * d.cW + cv_wi0[i];
    }
    // PIPE: physical chunk slot tid&7 holds logical chunk (tid&7) ^ swizzle(row); the swizzle (row>>1)&7 is the same
    // for all of a thread's rows (they are 32 apart), so the thread still owns ONE k chunk per tile.
    const int kchunk = PIPE ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7);
    const int k = blockIdx.y * k_tiles_per_split * BK + kchunk * VE;
    cv_c = k % d.cCin;
    const int t = k / d.cCin;
    cv_s = t % d.cKW;
    cv_r = t / d.cKW;
  }

  // global -> registers (VEC path).  Typed address_space(1) loads: a `cond ? *p : zero` on a generic pointer
  // makes the compiler select between the global address and a private zero and spill the staging registers.
  auto gload = [&](int kt) {
    if constexpr (VEC) {
      const int k0 = kt * BK;
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int c = tid + i * 256;
        ra[i] = (u32x4){0u, 0u, 0u, 0u};
        if constexpr (CONV) {
          const int hi = cv_hi0[i] + cv_r, wi = cv_wi0[i] + cv_s;
          if (cv_r < d.cKH && hi >= 0 && hi < d.cH && wi >= 0 && wi < d.cW)
            ra[i] = *(gptr_u4)(A + (cv_base[i] + (long)cv_r * d.cW + cv_s) * d.cCin + cv_c);
        } else if constexpr (AKC) {
          const int m = bm0 + (c >> 3), k = k0 + (c & 7) * VE;
          if (m < M && k < K) ra[i] = *(gptr_u4)(A + (long)m * lda + k);
        } else {
          constexpr int CPR = BM / VE;
          const int k = k0 + c / CPR, m = bm0 + (c % CPR) * VE;
          if (k < K && m < M) ra[i] = *(gptr_u4)(A + (long)k * lda + m);
        }
      }
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int c = tid + i * 256;
        rb[i] = (u32x4){0u, 0u, 0u, 0u};
        if constexpr (BKC) {
          const int n = bn0 + (c >> 3), k = k0 + (c & 7) * VE;
          if (n < N && k < K) rb[i] = *(gptr_u4)(B + (long)n * ldb + k);
        } else {
          constexpr int CPR = BN / VE;
          const int k = k0 + c / CPR, n = bn0 + (c % CPR) * VE;
          if (k < K && n < N) rb[i] = *(gptr_u4)(B + (long)k * ldb + n);
        }
      }
      if constexpr (CONV) {       // advance this thread's (r, s, c) by one K tile
        cv_c += BK;
        while (cv_c >= d.cCin) {
          cv_c -= d.cCin;
          if (++cv_s == d.cKW) { cv_s = 0; ++cv_r; }
        }
      }
    }
  };

  // registers (VEC) or global (scalar fallback) -> LDS buffer `buf`
  auto sstore = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * BUF_BYTES;
    unsigned char* sB = sA + A_BYTES;
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int c = tid + i * 256;
        if constexpr (AKC) {
          *(u32x4*)(sA + (c >> 3) * SA + (c & 7) * 16) = ra[i];
        } else {
          constexpr int CPR = BM / VE;
          *(u32x4*)(sA + (c / CPR) * SA + (c % CPR) * 16) = ra[i];
        }
      }
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int c = tid + i * 256;
        if constexpr (BKC) {
          *(u32x4*)(sB + (c >> 3) * SB + (c & 7) * 16) = rb[i];
        } else {
          constexpr int CPR = BN / VE;
          *(u32x4*)(sB + (c / CPR) * SB + (c % CPR) * 16) = rb[i];
        }
      }
    } else {
      const int k0 = kt * BK;
      for (int e = tid; e < BM * BK; e += 256) {
        TI v = (TI)0.f;
        if constexpr (AKC) {
          const int row = e / BK, kk = e % BK;
          const int m = bm0 + row, k = k0 + kk;
          if (m < M && k < K) v = A[(long)m * lda + k];
          *(TI*)(sA + row * SA + kk * SZ) = v;
        } else {
          const int kk = e / BM, row = e % BM;
          const int m = bm0 + row, k = k0 + kk;
          if (m < M && k < K) v = A[(long)k * lda + m];
          *(TI*)(sA + kk * SA + row * SZ) = v;
        }
      }
      for (int e = tid; e < BN * BK; e += 256) {
        TI v = (TI)0.f;
        if constexpr (BKC) {
          const int row = e / BK, kk = e % BK;
          const int n = bn0 + row, k = k0 + kk;
          if (n < N && k < K) v = B[(long)n * ldb + k];
          *(TI*)(sB + row * SB + kk * SZ) = v;
        } else {
          const int kk = e / BN, row = e % BN;
          const int n = bn0 + row, k = k0 + kk;
          if (n < N && k < K) v = B[(long)k * ldb + n];
          *(TI*)(sB + kk * SB + row * SZ) = v;
        }
      }
    }
  };

  // ---- PIPE: one K tile -> LDS stage `buf` by LDS-DMA.  Chunk c = tid + i*256 lands at byte 16*c of the stage's A (B)
  // image: the wave-instruction's 64 lanes write 1 KiB contiguously (wave-uniform base + lane*16).
  auto issue = [&](int kt, int buf) {
    if constexpr (PIPE) {
      unsigned char* sA = smem + buf * BUF_BYTES;
      unsigned char* sB = sA + A_BYTES;
      const int k0 = kt * BK;
      const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * VE;         // logical k offset of this thread's chunk
      const int wbase = (tid & ~63) * 16;
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const int c = tid + i * 256;
        const TI* src = (const TI*)g_zero16;
        if constexpr (CONV) {
          const int hi = cv_hi0[i] + cv_r, wi = cv_wi0[i] + cv_s;
          if (cv_r < d.cKH && hi >= 0 && hi < d.cH && wi >= 0 && wi < d.cW)
            src = (const TI*)d.A + (cv_base[i] + (long)cv_r * d.cW + cv_s) * d.cCin + cv_c;
        } else {
          const int m = bm0 + (c >> 3), k = k0 + kc;
          if (m < M && k < K) src = (const TI*)d.A + (long)m * lda + k;
        }
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(sA + i * 4096 + wbase), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        const int c = tid + i * 256;
        const TI* src = (const TI*)g_zero16;
        const int n = bn0 + (c >> 3), k = k0 + kc;
        if (n < N && k < K) src = (const TI*)d.B + (long)n * ldb + k;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)(sB + i * 4096 + wbase), 16, 0, 0);
      }
      if constexpr (CONV) {
        cv_c += BK;
        while (cv_c >= d.cCin) {
          cv_c -= d.cCin;
          if (++cv_s == d.cKW) { cv_s = 0; ++cv_r; }
        }
      }
    }
  };

  auto compute = [&](int buf) {
    const unsigned char* sA = smem + buf * BUF_BYTES;
    const unsigned char* sB = sA + A_BYTES;
    if constexpr (IS_BF16) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int r0 = wr * (BM / 2) + t * 16;
          if constexpr (PIPE) {
            fa[t] = *(const bf16x8*)(sA + (r0 + lr) * 128 + (((ks * 4 + lg) ^ (((r0 + lr) >> 1) & 7)) << 4));
          } else if constexpr (AKC) {
            fa[t] = *(const bf16x8*)(sA + (r0 + lr) * SA + ks * 64 + lg * 16);
          } else {
            // [k][m] image: lane (q=lr>>2, p=lr&3) addresses row k0+q, cols 4p..4p+3; receives column lr
            const unsigned char* p0 = sA + (ks * 32 + lg * 8 + (lr >> 2)) * SA + (r0 + 4 * (lr & 3)) * 2;
            bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
            bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * SA));
            fa[t] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int r0 = wc * (BN / 2) + t * 16;
          if constexpr (PIPE) {
            fb[t] = *(const bf16x8*)(sB + (r0 + lr) * 128 + (((ks * 4 + lg) ^ (((r0 + lr) >> 1) & 7)) << 4));
          } else if constexpr (BKC) {
            fb[t] = *(const bf16x8*)(sB + (r0 + lr) * SB + ks * 64 + lg * 16);
          } else {
            const unsigned char* p0 = sB + (ks * 32 + lg * 8 + (lr >> 2)) * SB + (r0 + 4 * (lr & 3)) * 2;
            bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
            bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * SB));
            fb[t] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // f32: lane group lg owns k = 8*lg .. 8*lg+7 of the 32-wide tile; MFMA step s
      // contracts k in {8g+s : g=0..3} (same map for A and B, so the sum covers all 32).
      float fa[TM][8], fb[TN][8];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int r0 = wr * (BM / 2) + t * 16;
        if constexpr (AKC) {
          const int sw = PIPE ? (((r0 + lr) >> 1) & 7) : 0;
          const float4 v0 = *(const float4*)(sA + (r0 + lr) * SA + (((2 * lg) ^ sw) << 4));
          const float4 v1 = *(const float4*)(sA + (r0 + lr) * SA + (((2 * lg + 1) ^ sw) << 4));
          fa[t][0] = v0.x; fa[t][1] = v0.y; fa[t][2] = v0.z; fa[t][3] = v0.w;
          fa[t][4] = v1.x; fa[t][5] = v1.y; fa[t][6] = v1.z; fa[t][7] = v1.w;
        } else {
#pragma unroll
          for (int s = 0; s < 8; ++s) fa[t][s] = *(const float*)(sA + (lg * 8 + s) * SA + (r0 + lr) * 4);
        }
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int r0 = wc * (BN / 2) + t * 16;
        if constexpr (BKC) {
          const int sw = PIPE ? (((r0 + lr) >> 1) & 7) : 0;
          const float4 v0 = *(const float4*)(sB + (r0 + lr) * SB + (((2 * lg) ^ sw) << 4));
          const float4 v1 = *(const float4*)(sB + (r0 + lr) * SB + (((2 * lg + 1) ^ sw) << 4));
          fb[t][0] = v0.x; fb[t][1] = v0.y; fb[t][2] = v0.z; fb[t][3] = v0.w;
          fb[t][4] = v1.x; fb[t][5] = v1.y; fb[t][6] = v1.z; fb[t][7] = v1.w;
        } else {
#pragma unroll
          for (int s = 0; s < 8; ++s) fb[t][s] = *(const float*)(sB + (lg * 8 + s) * SB + (r0 + lr) * 4);
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }
  };

  // ---- K loop over this block's split: [kt0, kt1)
  const int nk = (K + BK - 1) / BK;
  const int kt0 = blockIdx.y * k_tiles_per_split;
  const int kt1 = min(nk, kt0 + k_tiles_per_split);
  if constexpr (PIPE) {
    if (kt0 < kt1) {
      constexpr int NL = CA + CB;                       // LDS-DMA instructions per thread per stage
      STAMP(1);
      issue(kt0, 0);
      if (kt0 + 1 < kt1) issue(kt0 + 1, 1);
      int buf = 0;
      for (int kt = kt0; kt < kt1; ++kt) {
        // stage kt has landed once all but this wave's newest NL DMAs (stage kt+1) are done; the barrier then
        // publishes every wave's part of it and retires all reads of the stage that is about to be refilled.
        if (kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt == kt0) STAMP(2);
        if (kt + 2 < kt1 && !(DBG & 1)) issue(kt + 2, buf >= 1 ? buf - 1 : 2);      // (buf + 2) % 3
        if (!(DBG & 2)) compute(buf);
        buf = buf == 2 ? 0 : buf + 1;
      }
      __syncthreads();                                  // all fragment reads done before the epilogue reuses LDS
      STAMP(3);
    }
  } else if (kt0 < kt1) {
    int cur = 0;
    gload(kt0);
    sstore(kt0, 0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) gload(kt + 1);          // next tile's HBM/L2 loads fly under this tile's MFMAs
      compute(cur);
      if (more) sstore(kt + 1, cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue.  C/D map: col = lane&15, row = (lane>>4)*4 + reg
  TO* __restrict__ C = (TO*)d.C;
  const bool split = gridDim.y > 1;

  // Staged path (conv / plain products that overwrite C): the tile goes through LDS so that every global store
  // is a 16-byte piece of a contiguous output row (a direct store of the MFMA layout writes 32-byte row
  // fragments, 2 bytes per lane), and the BatchNorm column sums are folded across the block before ONE atomic
  // per column lands in one of `stats_nrep` replicas (few adders per address: global f32 atomics serialise).
  if constexpr (EPI != EPI_HIGHWAY) {
    constexpr int OSZ = sizeof(TO);
    constexpr int OVE = 16 / OSZ;
    constexpr int SC = BN * OSZ + 16;                       // LDS row stride of the C tile
    static_assert(BM * SC + 4 * (BN / 2) * 2 * 4 <= SMEM_BYTES, "C tile + stats scratch must fit the LDS allocation");
    const bool staged = !split && !d.accumulate && (N % OVE == 0) && (d.ldc % OVE == 0) && ((((uintptr_t)d.C) & 15) == 0);
    if (staged) {
      unsigned char* sC = smem;
      float* sStat = (float*)(smem + BM * SC);               // [4 waves][BN/2][2]
      float st_s[TN], st_q[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nl = wc * (BN / 2) + j * 16 + lr;
        const int n = bn0 + nl;
        const float bias = (d.bias && n < N) ? d.bias[n] : 0.f;
        st_s[j] = 0.f; st_q[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ml = wr * (BM / 2) + i * 16 + lg * 4 + r;
            const float v = d.alpha * acc[i][j][r] + bias;
            *(TO*)(sC + ml * SC + nl * OSZ) = from_f32<TO>(v);
            if (EPI == EPI_BNSTATS && bm0 + ml < M) { st_s[j] += v; st_q[j] += v * v; }
          }
        }
        if constexpr (EPI == EPI_BNSTATS) {
          st_s[j] += __shfl_xor(st_s[j], 16, 64); st_q[j] += __shfl_xor(st_q[j], 16, 64);
          st_s[j] += __shfl_xor(st_s[j], 32, 64); st_q[j] += __shfl_xor(st_q[j], 32, 64);
          if (lg == 0) { sStat[(w * (BN / 2) + j * 16 + lr) * 2] = st_s[j]; sStat[(w * (BN / 2) + j * 16 + lr) * 2 + 1] = st_q[j]; }
        }
      }
      __syncthreads();
      STAMP(4);
      if constexpr (EPI == EPI_BNSTATS) {
        if (tid < BN) {                                        // column tid: waves (wr=0, wc) and (wr=1, wc)
          const int cwc = tid / (BN / 2), cl = tid % (BN / 2), n = bn0 + tid;
          if (n < N) {
            const float s0 = sStat[((0 * 2 + cwc) * (BN / 2) + cl) * 2] + sStat[((1 * 2 + cwc) * (BN / 2) + cl) * 2];
            const float q0 = sStat[((0 * 2 + cwc) * (BN / 2) + cl) * 2 + 1] + sStat[((1 * 2 + cwc) * (BN / 2) + cl) * 2 + 1];
            float* st = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * N;
            atomicAdd(&st[n], s0);
            atomicAdd(&st[N + n], q0);
          }
        }
      }
      constexpr int CPR = BN / OVE;                            // 16-B chunks per tile row
      for (int c = tid; c < BM * CPR; c += 256) {
        const int ml = c / CPR, cc = c % CPR;
        const int m = bm0 + ml, n = bn0 + cc * OVE;
        if (m < M && n < N) *(u32x4*)(C + (long)m * d.ldc + n) = *(const u32x4*)(sC + ml * SC + cc * 16);
      }
      STAMP(5);
      return;
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = bn0 + wc * (BN / 2) + j * 16 + lr;
      if (n >= N) continue;
      const float bias = (d.bias && blockIdx.y == 0) ? d.bias[n] : 0.f;
      const int mb = bm0 + wr * (BM / 2) + i * 16 + lg * 4;
      float keep4[4] = {1.f, 1.f, 1.f, 1.f};
      if constexpr (EPI == EPI_HIGHWAY) {
        if (!d.mask && d.use_philox) {       // one Philox4x32 call serves the lane's 4 consecutive rows
          uint32_t r0, r1, r2, r3;
          Philox::gen4(d.seed_dev ? *d.seed_dev : d.seed, d.stream, (uint64_t)(mb >> 2) * (uint64_t)N + (uint64_t)n, r0, r1, r2, r3);
          keep4[0] = Philox::u01(r0) >= d.drop_p ? 1.f : 0.f;
          keep4[1] = Philox::u01(r1) >= d.drop_p ? 1.f : 0.f;
          keep4[2] = Philox::u01(r2) >= d.drop_p ? 1.f : 0.f;
          keep4[3] = Philox::u01(r3) >= d.drop_p ? 1.f : 0.f;
        }
      }
      float st_s = 0.f, st_q = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mb + r;
        if (m >= M) continue;
        float v = d.alpha * acc[i][j][r] + bias;
        if constexpr (EPI == EPI_BNSTATS) {
          C[(long)m * d.ldc + n] = from_f32<TO>(v);
          st_s += v; st_q += v * v;
        } else if constexpr (EPI == EPI_PLAIN) {
          const long o = (long)m * d.ldc + n;
          if constexpr (sizeof(TO) == 4) {
            if (split) { atomicAdd((float*)&C[o], v); continue; }
          }
          if (d.accumulate) v += to_f32<TO>(C[o]);
          C[o] = from_f32<TO>(v);
        } else {   // EPI_HIGHWAY
          const float x = to_f32<TI>(((const TI*)d.X)[(long)m * d.ldx + n]);
          const float sg = 1.f / (1.f + expf(-v));
          const float y = sg * fmaxf(v, 0.f) + (1.f - sg) * x;
          if (d.Hpre) d.Hpre[(long)m * d.ldh + n] = v;          // (null: forward only, nothing saved for a backward pass)
          float keep = keep4[r];
          if (d.mask) keep = (float)d.mask[(long)m * d.ldmask + n];
          if (d.mask_out) d.mask_out[(long)m * d.ldmask_out + n] = (uint8_t)keep;
          C[(long)m * d.ldc + n] = from_f32<TO>(y * keep * d.keep_scale);
        }
      }
      if constexpr (EPI == EPI_BNSTATS) {
        // column n lives in lanes {lr, lr+16, lr+32, lr+48}: fold the 4 row groups, one atomic pair per column
        st_s += __shfl_xor(st_s, 16, 64); st_q += __shfl_xor(st_q, 16, 64);
        st_s += __shfl_xor(st_s, 32, 64); st_q += __shfl_xor(st_q, 32, 64);
        float* st = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * N;
        if (lg == 0) { atomicAdd(&st[n], st_s); atomicAdd(&st[N + n], st_q); }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// tile8: the bf16 k-contiguous product / implicit-GEMM convolution with EIGHT waves per 128-row tile (2 per SIMD).
//
// Measured on MI355X (tools/gemm_stamps.py) the 4-wave kernel above spends ~1500 cycles per 64-deep K tile where the MFMA
// work is 512: one wave per SIMD serialises its LDS-DMA issue, its fragment reads and its MFMAs, and two tiles in flight do
// not cover the ~2000-cycle loaded DMA latency.  Here a SIMD holds two waves (one can issue DMA / wait on LDS while the
// other feeds the matrix pipe), the ring is NS deep (NS-1 tiles in flight) and addressing is cheap:
//   * tiles reach LDS by `buffer_load_dwordx4 ... lds` behind ONE buffer descriptor per operand: a 32-bit byte offset per
//     chunk, and an offset beyond the descriptor's extent (conv padding, M/N/K tails, tiles past the end of K) reads as
//     zero without touching memory -- no 64-bit address math, no branches, no zero page;
//   * every iteration issues exactly NL DMAs, so the counted s_waitcnt is a constant and the loop body is branch-free.
// Waves are 4 (M) x 2 (N): a wave owns 32 x BN/2 of the tile.  LDS image, swizzle and C/D layout as in the kernel above.
template <typename TO, int BN, int EPI, bool CONV, int NS, bool ABN = false, bool ARES = false, int AMAXK = (ARES ? 2048 : 1024)>
__global__ __launch_bounds__(512) void tile8_kernel(const GemmDesc d, const unsigned a_bytes, const unsigned b_bytes) {
  static_assert(!ARES || ABN, "the residual-on-load form extends the A-side BatchNorm");
  constexpr int BM = 128, BK = 64, NT = 512;
  constexpr int TM = 2, TN = BN / 32;
  constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;          // 16-B chunks per thread per tile
  constexpr int NL = CA + CB + (ARES ? CA : 0);              // LDS-DMA instructions per thread per stage
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, R_BYTES = ARES ? A_BYTES : 0, STAGE = A_BYTES + B_BYTES + R_BYTES;
  constexpr int OSZ = sizeof(TO), OVE = 16 / OSZ, SC = BN * OSZ + 16;
  constexpr int EPI_BYTES = EPI == EPI_HIGHWAY ? BM * (BN + 4) * 4 : (EPI == EPI_GUMBELMAX ? 8 * BN * 4 : BM * SC + 8 * (BN / 2) * 2 * 4);   // highway stages h in f32
  constexpr int RING_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
  constexpr int ABN_MAXK = AMAXK;                             // A-side BatchNorm: [scale, shift] per input channel behind the ring
  constexpr int SMEM_BYTES = RING_BYTES + (ABN ? ABN_MAXK * 8 : 0) + (ARES ? ABN_MAXK * 8 : 0);     // ARES: + the shortcut's [scale, shift]
  constexpr unsigned OOB = 0x80000000u;                       // >= any extent this kernel is launched with
  static_assert(CB >= 1 && SMEM_BYTES <= 160 * 1024, "tile8 LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  STAMP(0);
  const int wr = w >> 1, wc = w & 1;
  const int lr = lane & 15, lg = lane >> 4;

  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int bid = xcd_run(blockIdx.x, gridDim.x);
  const int bm0 = (d.n_fast ? bid / tiles_n : bid % tiles_m) * BM;
  const int bn0 = (d.n_fast ? bid % tiles_n : bid / tiles_m) * BN;
  const int M = d.M, N = d.N, K = d.K;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, (int)b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(ARES ? d.res : d.A), 0, (int)a_bytes, 0x00020000);

  // ---- per-thread chunk coordinates: rows (tid>>3) + 64 i, physical chunk slot tid&7 = logical chunk ^ swizzle(row)
  const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;           // logical k offset (elements) inside a K tile
  int a_off[CA], a_hi0[CA], a_wi0[CA];
  bool a_ok[CA];
  int cv_r = 0, cv_s = 0, cv_c = 0;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int m = bm0 + (tid >> 3) + i * 64;
    a_ok[i] = m < M;
    if constexpr (CONV) {
      const int mm = a_ok[i] ? m : 0;
      const int wo = mm % d.cWo, t = mm / d.cWo;
      const int ho = t % d.cHo, n = t / d.cHo;
      a_hi0[i] = a_ok[i] ? ho * d.cStride - d.cPad : -(1 << 28);      // rows past M never validate
      a_wi0[i] = wo * d.cStride - d.cPad;
      a_off[i] = ((n * d.cH + a_hi0[i]) * d.cW + a_wi0[i]) * d.cCin;
    } else {
      a_hi0[i] = a_wi0[i] = 0;
      a_off[i] = m * (int)d.lda;
    }
  }
  if constexpr (CONV) {
    cv_c = kc % d.cCin;
    const int t = kc / d.cCin;
    cv_s = t % d.cKW;
    cv_r = t / d.cKW;
  }
  int b_off[CB];
  bool b_ok[CB];
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int n = bn0 + (tid >> 3) + i * 64;
    b_ok[i] = n < N;
    b_off[i] = n * (int)d.ldb;
  }
  const int wbase = (tid & ~63) * 16;

  // one K tile -> ring stage `st`; called once per kt in increasing order (the conv (r,s,c) runs along)
  auto issue = [&](int kt, int st) {
    unsigned char* sA = smem + st * STAGE;
    unsigned char* sB = sA + A_BYTES;
    const int k = kt * BK + kc;
    const bool kok = k < K;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      unsigned voff;
      if constexpr (CONV) {
        const bool ok = (unsigned)(a_hi0[i] + cv_r) < (unsigned)d.cH & (unsigned)(a_wi0[i] + cv_s) < (unsigned)d.cW & kok;
        voff = ok ? (unsigned)(a_off[i] + (cv_r * d.cW + cv_s) * d.cCin + cv_c) * 2u : OOB;
      } else {
        voff = (a_ok[i] & kok) ? (unsigned)(a_off[i] + k) * 2u : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(sA + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, CONV ? GIC_TRUNK_NT : 0);
      if constexpr (ARES)        // the shortcut tile: same rows and channels of `res`, behind the B tile of the stage
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (lds_void_ptr)(sB + B_BYTES + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, CONV ? GIC_TRUNK_NT : 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const unsigned voff = (b_ok[i] & kok) ? (unsigned)(b_off[i] + k) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(sB + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, 0);
    }
    if constexpr (CONV) {
      cv_c += BK;
      while (cv_c >= d.cCin) {
        cv_c -= d.cCin;
        if (++cv_s == d.cKW) { cv_s = 0; ++cv_r; }
      }
    }
  };

  // ---- A-side BatchNorm + ReLU: per-channel [scale, shift] from the producer's sums, then each thread normalises the chunks it
  // DMA'd itself (its own vmcnt wait covers them) in LDS before the barrier publishes the stage.
  float* coef = (float*)(smem + RING_BYTES);
  if constexpr (ABN) {
    const int Cn = CONV ? d.cCin : K;                          // channels of the normalised operand
    for (int c = tid; c < ABN_MAXK; c += NT) {
      float sc = 0.f, sh = 0.f;
      if (c < Cn) {
        float s1, s2;
        fold_replicas(d.in_stats, d.in_nrep, Cn, c, s1, s2);
        const float mean = s1 * d.in_inv_count;
        const float var = fmaxf(s2 * d.in_inv_count - mean * mean, 0.f);
        sc = d.in_gamma[c] * rsqrtf(var + 1e-5f);              // kBnEps of encoder.hip (nn.BatchNorm2d default)
        sh = d.in_beta[c] - mean * sc;
      }
      coef[2 * c] = sc; coef[2 * c + 1] = sh;
      if constexpr (ARES) {      // the shortcut: its own BatchNorm (projection) or identity (scale 1, shift 0)
        float rs = 1.f, rt = 0.f;
        if (d.res_stats && c < Cn) {
          float s1, s2;
          fold_replicas(d.res_stats, d.res_nrep, Cn, c, s1, s2);
          const float mean = s1 * d.res_inv_count;
          const float var = fmaxf(s2 * d.res_inv_count - mean * mean, 0.f);
          rs = d.res_gamma[c] * rsqrtf(var + 1e-5f);
          rt = d.res_beta[c] - mean * rs;
        }
        coef[2 * ABN_MAXK + 2 * c] = rs; coef[2 * ABN_MAXK + 2 * c + 1] = rt;
      }
    }
    __syncthreads();
  }
  const bool wb = ARES && d.out_wb && bn0 == 0;              // first N tile: also materialise the formed operand (the block output)
  auto abn = [&](int kt, int st) {
    if constexpr (ABN) {
      const int k = kt * BK + kc;
      if (k < K) {
        // channel and tap of this thread's chunks (all CA of them share k): a 16-byte chunk never straddles a tap (Cin % 8 == 0)
        const int ch = CONV ? k % d.cCin : k;
        const int tap = CONV ? k / d.cCin : 0;
        const int ts = tap % (CONV ? d.cKW : 1), tr = tap / (CONV ? d.cKW : 1);
        const float4* cp = (const float4*)(coef + 2 * ch);
        const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
        const float scl[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
        const float sft[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
        float rsc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, rsf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (ARES) {
          const float4* rp = (const float4*)(coef + 2 * ABN_MAXK + 2 * ch);
          const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
          rsc[0] = r0.x; rsc[1] = r0.z; rsc[2] = r1.x; rsc[3] = r1.z; rsc[4] = r2.x; rsc[5] = r2.z; rsc[6] = r3.x; rsc[7] = r3.z;
          rsf[0] = r0.y; rsf[1] = r0.w; rsf[2] = r1.y; rsf[3] = r1.w; rsf[4] = r2.y; rsf[5] = r2.w; rsf[6] = r3.y; rsf[7] = r3.w;
        }
#pragma unroll
        for (int i = 0; i < CA; ++i) {
          // padding taps and rows past M were zero-filled by the DMA and stay zero (the reference pads the NORMALISED tensor)
          const bool ok = CONV ? ((unsigned)(a_hi0[i] + tr) < (unsigned)d.cH & (unsigned)(a_wi0[i] + ts) < (unsigned)d.cW) : a_ok[i];
          if (!ok) continue;
          bf16x8* p = (bf16x8*)(smem + st * STAGE + i * (NT * 16) + tid * 16);
          bf16x8 v = *p;
          if constexpr (ARES) {
            const bf16x8 r = *(const bf16x8*)(smem + st * STAGE + A_BYTES + B_BYTES + i * (NT * 16) + tid * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)fmaxf((float)v[e] * scl[e] + sft[e] + ((float)r[e] * rsc[e] + rsf[e]), 0.f);
            // 1x1 / stride 1 / pad 0: operand row m, channels k..k+7 = element a_off[i] + k of the [M, Cin] block output
            if (wb) *(bf16x8*)((bf16_t*)d.out_wb + (long)a_off[i] + k) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)fmaxf((float)v[e] * scl[e] + sft[e], 0.f);
          }
          *p = v;
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int st) {
    const unsigned char* sA = smem + st * STAGE;
    const unsigned char* sB = sA + A_BYTES;
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = wr * 32 + t * 16 + lr;
        fa[ks][t] = *(const bf16x8*)(sA + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int row = wc * (BN / 2) + t * 16 + lr;
        fb[ks][t] = *(const bf16x8*)(sB + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
  };

  // fragment reads / MFMAs of one 32-deep half (ks) of a K tile, as separate phases for the software pipeline below
  auto read_half = [&](int st, int ks, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
    const unsigned char* sA = smem + st * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = wr * 32 + t * 16 + lr;
      fa[t] = *(const bf16x8*)(sA + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
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

  // ---- K loop: NS-1 tiles in flight.  Tiles past the end of K are all-OOB DMAs (zero fill, no traffic), so the count
  // of outstanding DMAs is the same in every iteration.
  const int nk = (K + BK - 1) / BK;
  STAMP(1);
  if constexpr (NS == 1) {
    // shallow K (one or two tiles): no ring; the LDS footprint is the C tile's, so up to four blocks share a CU and hide
    // each other's load latency and epilogue
    for (int kt = 0; kt < nk; ++kt) {
      if (kt) __builtin_amdgcn_s_barrier();            // every wave is done reading the previous tile
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      abn(kt, 0);
      __builtin_amdgcn_s_barrier();
      compute(0);
    }
  } else if constexpr (NS < 4) {
    // 2-stage ring: two workgroups share the CU and cover each other's phases; the plain loop measures faster here
    issue(0, 0);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      abn(kt, st);
      __builtin_amdgcn_s_barrier();                      // stage kt published; every wave is done reading stage kt-1
      if (kt == 0) STAMP(2);
      if (!(DBG & 1)) issue(kt + 1, st ^ 1);
      if (!(DBG & 2)) compute(st);
      st ^= 1;
    }
  } else {
    // Software pipeline across K tiles: the second half's fragments of tile kt-1 stay in registers over the barrier, so that
    // every MFMA phase has the next phase's LDS reads in flight under it (all 8 waves leave the barrier in lockstep: without
    // this the CU alternates between an LDS-read phase and an MFMA phase):
    //     barrier | read A(kt) | mfma B(kt-1) | DMA(kt+NS-1) | read B(kt) | mfma A(kt)       A / B = 32-deep halves
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);
    bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    int st = 0, st_fill = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
      // stage kt has landed once all but this wave's newest (NS-2) tiles are done.  This wave's reads of stage kt-1 (their
      // data is needed below anyway) must have returned before the barrier lets anyone refill that stage.
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NS >= 2 ? (NS - 2) * NL : 0) : "memory");
      abn(kt, st);
      __builtin_amdgcn_s_barrier();
      if (kt == 0) STAMP(2);
      if (!(DBG & 2)) {
        read_half(st, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt > 0) mfma_half(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(DBG & 1)) issue(kt + NS - 1, st_fill);
      if (!(DBG & 2)) {
        __builtin_amdgcn_sched_barrier(0);
        read_half(st, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa0, fb0);
      }
      st = st == NS - 1 ? 0 : st + 1;
      st_fill = st_fill == NS - 1 ? 0 : st_fill + 1;
    }
    if (!(DBG & 2) && nk > 0) mfma_half(fa1, fb1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stray zero-fill DMAs must not land in the C tile
  __syncthreads();
  STAMP(3);

  if constexpr (EPI == EPI_GUMBELMAX) {
    // rows m = vocabulary entries, columns n = roll-out rows.  A lane's four accumulator registers of a 16x16 block are four CONSECUTIVE
    // vocabulary entries of one roll-out row: one Philox4x32 call (or one 16-byte load of explicit uniforms) and one bias load per block.
    float* sV = (float*)smem;                    // [4 (wr)][BN] best value
    int* sI = (int*)(smem + 4 * BN * 4);         // [4 (wr)][BN] its vocabulary index
    const float eps = 1e-10f;                    // generator.py:84
    const uint64_t seed = d.seed_dev ? *d.seed_dev : d.seed;
    const float T = d.gm_temperature;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wc * (BN / 2) + j * 16 + lr;
      const int n = bn0 + nl;
      float best = -INFINITY;
      int best_i = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m0 = bm0 + wr * 32 + i * 16 + lg * 4;
        if (m0 < M && n < N) {                   // (M % 4 == 0: a quad is inside the vocabulary or past it as a whole)
          const float4 bv = *(const float4*)(d.gm_bias + m0);
          const float bia[4] = {bv.x, bv.y, bv.z, bv.w};
          float uu[4];
          if (d.gm_u) {
            const float4 uv = *(const float4*)(d.gm_u + (long)n * d.gm_ldu + m0);
            uu[0] = uv.x; uu[1] = uv.y; uu[2] = uv.z; uu[3] = uv.w;
          } else {
            uint32_t r0, r1, r2, r3;
            Philox::gen4(seed, d.stream, (uint64_t)n * (uint64_t)(M >> 2) + (uint64_t)(m0 >> 2), r0, r1, r2, r3);
            uu[0] = Philox::u01(r0); uu[1] = Philox::u01(r1); uu[2] = Philox::u01(r2); uu[3] = Philox::u01(r3);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = -__logf(-__logf(uu[r] + eps) + eps);      // (bf16 compute mode: the hardware log, as the unfused kernel)
            const float y = (acc[i][j][r] + bia[r] + g) * T;
            if (y > best) { best = y; best_i = m0 + r; }
          }
        }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {       // the four lane groups hold other vocabulary rows of the same roll-out row
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
      }
      if (lg == 0) { sV[wr * BN + nl] = best; sI[wr * BN + nl] = best_i; }
    }
    __syncthreads();
    if (tid < BN && bn0 + tid < N) {
      float best = sV[tid];
      int best_i = sI[tid];
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        const float ob = sV[r * BN + tid];
        const int oi = sI[r * BN + tid];
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
      }
      if (best_i != 0x7fffffff) atomicMax(d.gm_rowkey + bn0 + tid, row_key(best, best_i));
    }
    STAMP(5);
    return;
  }
  TO* __restrict__ C = (TO*)d.C;
  if constexpr (EPI == EPI_HIGHWAY) {
    // h = acc + bias (saved); y = sig(h) relu(h) + (1 - sig(h)) x; C = y * keep * keep_scale.
    // Row-padded operands (every leading dimension covers whole 8-column groups, as the discriminator's Fp-padded buffers do):
    // h goes through LDS and each thread owns a 4-row x 8-column patch = one Philox draw per column (4 rows each), 16-byte
    // accesses to X / Hpre / C and 8-byte ones to the keep mask.  Pad columns of C are written as zero.
    const int n8 = (N + 7) & ~7;
    const bool wide = sizeof(TO) == 2 && d.ldc >= n8 && d.ldx >= n8 && d.ldh >= n8 && (!d.mask_out || d.ldmask_out >= n8) &&
                      d.ldc % 8 == 0 && d.ldx % 8 == 0 && d.ldh % 4 == 0 && (!d.mask_out || d.ldmask_out % 8 == 0) &&
                      ((((uintptr_t)d.C) | ((uintptr_t)d.X) | ((uintptr_t)d.Hpre)) & 15) == 0 && (((uintptr_t)d.mask_out) & 7) == 0;
    if (wide) {
      constexpr int SH = BN + 4;
      float* sH = (float*)smem;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nl = wc * (BN / 2) + j * 16 + lr;
        const float bias = (d.bias && bn0 + nl < N) ? d.bias[bn0 + nl] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) sH[(wr * 32 + i * 16 + lg * 4 + r) * SH + nl] = d.alpha * acc[i][j][r] + bias;
      }
      __syncthreads();
      constexpr int CG = BN / 8;
      const int pc = tid % CG, pr = tid / CG;
      const int n0 = bn0 + pc * 8, m0 = bm0 + pr * 4;
      if (pr < BM / 4 && n0 < N && m0 < M) {
        float keep[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          keep[0][e] = keep[1][e] = keep[2][e] = keep[3][e] = 1.f;
          if (!d.mask && d.use_philox && n0 + e < N) {
            uint32_t r0, r1, r2, r3;
            Philox::gen4(d.seed_dev ? *d.seed_dev : d.seed, d.stream, (uint64_t)(m0 >> 2) * (uint64_t)N + (uint64_t)(n0 + e), r0, r1, r2, r3);
            keep[0][e] = Philox::u01(r0) >= d.drop_p ? 1.f : 0.f;
            keep[1][e] = Philox::u01(r1) >= d.drop_p ? 1.f : 0.f;
            keep[2][e] = Philox::u01(r2) >= d.drop_p ? 1.f : 0.f;
            keep[3][e] = Philox::u01(r3) >= d.drop_p ? 1.f : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + r;
          if (m >= M) break;
          const float4 h0 = *(const float4*)(sH + (pr * 4 + r) * SH + pc * 8), h1 = *(const float4*)(sH + (pr * 4 + r) * SH + pc * 8 + 4);
          const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
          const bf16x8 xv = *(const bf16x8*)((const bf16_t*)d.X + (long)m * d.ldx + n0);
          bf16x8 yv;
          unsigned long long kb = 0ull;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool live = n0 + e < N;
            const float sg = 1.f / (1.f + expf(-h[e]));
            const float y = sg * fmaxf(h[e], 0.f) + (1.f - sg) * (float)xv[e];
            float k = keep[r][e];
            if (d.mask && live) k = (float)d.mask[(long)m * d.ldmask + n0 + e];
            kb |= (unsigned long long)(live ? (unsigned)k : 0u) << (8 * e);
            yv[e] = (bf16_t)(live ? y * k * d.keep_scale : 0.f);
          }
          if (d.Hpre) {
            *(float4*)(d.Hpre + (long)m * d.ldh + n0) = h0;
            *(float4*)(d.Hpre + (long)m * d.ldh + n0 + 4) = h1;
          }
          if (d.mask_out) *(unsigned long long*)(d.mask_out + (long)m * d.ldmask_out + n0) = kb;
          *(bf16x8*)((bf16_t*)C + (long)m * d.ldc + n0) = yv;
        }
      }
      STAMP(5);
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = bn0 + wc * (BN / 2) + j * 16 + lr;
        if (n >= N) continue;
        const float bias = d.bias ? d.bias[n] : 0.f;
        const int mb = bm0 + wr * 32 + i * 16 + lg * 4;
        float keep4[4] = {1.f, 1.f, 1.f, 1.f};
        if (!d.mask && d.use_philox) {       // one Philox4x32 call serves the lane's 4 consecutive rows
          uint32_t r0, r1, r2, r3;
          Philox::gen4(d.seed_dev ? *d.seed_dev : d.seed, d.stream, (uint64_t)(mb >> 2) * (uint64_t)N + (uint64_t)n, r0, r1, r2, r3);
          keep4[0] = Philox::u01(r0) >= d.drop_p ? 1.f : 0.f;
          keep4[1] = Philox::u01(r1) >= d.drop_p ? 1.f : 0.f;
          keep4[2] = Philox::u01(r2) >= d.drop_p ? 1.f : 0.f;
          keep4[3] = Philox::u01(r3) >= d.drop_p ? 1.f : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mb + r;
          if (m >= M) continue;
          const float v = d.alpha * acc[i][j][r] + bias;
          const float x = to_f32<bf16_t>(((const bf16_t*)d.X)[(long)m * d.ldx + n]);
          const float sg = 1.f / (1.f + expf(-v));
          const float y = sg * fmaxf(v, 0.f) + (1.f - sg) * x;
          if (d.Hpre) d.Hpre[(long)m * d.ldh + n] = v;          // (null: forward only, nothing saved for a backward pass)
          float keep = keep4[r];
          if (d.mask) keep = (float)d.mask[(long)m * d.ldmask + n];
          if (d.mask_out) d.mask_out[(long)m * d.ldmask_out + n] = (uint8_t)keep;
          C[(long)m * d.ldc + n] = from_f32<TO>(y * keep * d.keep_scale);
        }
      }
    }
    STAMP(5);
    return;
  }
  // ---- epilogue: C tile through LDS (16-byte row stores), BatchNorm column sums folded across the block
  unsigned char* sC = smem;
  float* sStat = (float*)(smem + BM * SC);                   // [8 waves][BN/2][2]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wc * (BN / 2) + j * 16 + lr;
    const int n = bn0 + nl;
    const float bias = (d.bias && n < N) ? d.bias[n] : 0.f;
    float st_s = 0.f, st_q = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ml = wr * 32 + i * 16 + lg * 4 + r;
        const float v = d.alpha * acc[i][j][r] + bias;
        *(TO*)(sC + ml * SC + nl * OSZ) = from_f32<TO>(v);
        if (EPI == EPI_BNSTATS && bm0 + ml < M) { st_s += v; st_q += v * v; }
      }
    }
    if constexpr (EPI == EPI_BNSTATS) {
      st_s += __shfl_xor(st_s, 16, 64); st_q += __shfl_xor(st_q, 16, 64);
      st_s += __shfl_xor(st_s, 32, 64); st_q += __shfl_xor(st_q, 32, 64);
      if (lg == 0) { sStat[(w * (BN / 2) + j * 16 + lr) * 2] = st_s; sStat[(w * (BN / 2) + j * 16 + lr) * 2 + 1] = st_q; }
    }
  }
  __syncthreads();
  STAMP(4);
  if constexpr (EPI == EPI_BNSTATS) {
    if (tid < BN) {                                          // column tid: waves (wr = 0..3, wc)
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
  }
  constexpr int CPR = BN / OVE;                              // 16-B chunks per tile row
  for (int c = tid; c < BM * CPR; c += NT) {
    const int ml = c / CPR, cc = c % CPR;
    const int m = bm0 + ml, n = bn0 + cc * OVE;
    if (m < M && n < N) {
      u32x4 v = *(const u32x4*)(sC + ml * SC + cc * 16);
      if (d.accumulate) {                                    // C += result (f32 gradients accumulated across passes)
        const u32x4 o = *(const u32x4*)(C + (long)m * d.ldc + n);
        TO* pv = (TO*)&v;
        const TO* po = (const TO*)&o;
#pragma unroll
        for (int e = 0; e < OVE; ++e) pv[e] = from_f32<TO>(to_f32<TO>(pv[e]) + to_f32<TO>(po[e]));
      }
      *(u32x4*)(C + (long)m * d.ldc + n) = v;
    }
  }
  STAMP(5);
}

// Launch the 8-wave kernel if the product qualifies (bf16 k-contiguous operands under 2 GiB each, a plain / BatchNorm-sum
// epilogue that overwrites a 16-byte-aligned C); returns false to fall back to gemm_kernel.
template <typename TO, int EPI, bool CONV>
bool try_tile8(const GemmDesc& d_in, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_TILE8") != nullptr;
  GemmDesc d = d_in;
  constexpr int OVE = 16 / (int)sizeof(TO);
  if (off || d.M < 128) return false;
  if (EPI != EPI_HIGHWAY && ((d.N % OVE) || (d.ldc % OVE) || (((uintptr_t)d.C) & 15))) return false;
  if (EPI == EPI_HIGHWAY && d.in_dtype != DT_BF16) return false;
  // a plain product must fill the chip with 128-row tiles and be deep enough to amortise the ring (else: gemm_kernel, split-K)
  if (!CONV && ((long)cdiv(d.M, 128) * cdiv(d.N, 64) < 128 || d.K < 256)) return false;
  const long a_elems = CONV ? (long)(d.M / (d.cHo * d.cWo)) * d.cH * d.cW * d.cCin : (long)(d.M - 1) * d.lda + d.K;
  const long b_elems = (long)(d.N - 1) * d.ldb + d.K;
  if (a_elems * 2 >= (1l << 31) || b_elems * 2 >= (1l << 31)) return false;
  const long big_tiles = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
  static const int big_min = [] { const char* e = getenv("GIC_TILE8_BIG_MIN"); return e ? atoi(e) : 160; }();
  static const int ns2_tiles = [] { const char* e = getenv("GIC_TILE8_NS2_TILES"); return e ? atoi(e) : 256; }();
  static const int min_nk = [] { const char* e = getenv("GIC_TILE8_MIN_NK"); return e ? atoi(e) : 1; }();
  if (cdiv(d.K, 64) < min_nk) return false;
  const unsigned ab = (unsigned)(a_elems * 2), bb = (unsigned)(b_elems * 2);
  // what an XCD's concurrent workgroups share (common.h): A bytes actually gathered (a strided 1x1 touches 1 / stride^2 of its input)
  d.n_fast = xcd_share_a(2 * (a_elems < (long)d.M * d.K ? a_elems : (long)d.M * d.K), 2l * d.N * d.K, cdiv(d.N, d.N >= 128 ? 128 : 64));
  // deep ring (4 stages, 128 KB: one block per CU) when the grid is about one block per CU; with several blocks per CU a
  // 2-stage ring (64 KB) lets two blocks share the CU so one block's epilogue runs under the other's K loop
  if constexpr (EPI == EPI_BNSTATS && CONV) {
    if (d.in_stats && d.res) {      // + residual on load: 1x1 / stride 1 / pad 0 only, one- or two-stage ring (two tiles per stage on the A side)
      if (d.cKH != 1 || d.cKW != 1 || d.cStride != 1 || d.cPad != 0 || d.cCin % 8 || d.cCin > 2048 || !d.in_gamma || !d.in_beta ||
          d.in_inv_count <= 0.f || (d.res_stats && (!d.res_gamma || !d.res_beta || d.res_inv_count <= 0.f)))
        return false;
      const bool n128 = d.N >= 128;
      const long tiles = n128 ? big_tiles : (long)cdiv(d.M, 128) * cdiv(d.N, 64);
      const dim3 grid((unsigned)tiles), block(512);
      // ring-less (one stage: two A-side tiles + B): 2-3 workgroups share a CU and cover each other's DMA latency and LDS rewrite;
      // the coefficient table is sized by the channel count (512 -> 8 KB, 2048 -> 32 KB)
      static const int res_ns = [] { const char* e = getenv("GIC_TILE8_RES_NS"); return e ? atoi(e) : 1; }();
      if (d.cCin <= 512) {
        if (res_ns == 1) {
          if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 1, true, true, 512>), grid, block, 0, stream, d, ab, bb);
          else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 1, true, true, 512>), grid, block, 0, stream, d, ab, bb);
        } else {
          if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 2, true, true, 512>), grid, block, 0, stream, d, ab, bb);
          else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 2, true, true, 512>), grid, block, 0, stream, d, ab, bb);
        }
      } else {
        if (res_ns == 1) {
          if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 1, true, true, 2048>), grid, block, 0, stream, d, ab, bb);
          else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 1, true, true, 2048>), grid, block, 0, stream, d, ab, bb);
        } else {
          if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 2, true, true, 2048>), grid, block, 0, stream, d, ab, bb);
          else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 2, true, true, 2048>), grid, block, 0, stream, d, ab, bb);
        }
      }
      return true;
    }
    if (d.in_stats) {      // A-side BatchNorm + ReLU: whole 16-byte chunks per tap, channels within the LDS table
      if (d.cCin % 8 || d.cCin > 1024 || !d.in_gamma || !d.in_beta || d.in_inv_count <= 0.f) return false;
      const int nk = cdiv(d.K, 64);
      const bool n128 = d.N >= 128 && big_tiles >= big_min;
      const long tiles = n128 ? big_tiles : (long)cdiv(d.M, 128) * cdiv(d.N, 64);
      const dim3 grid((unsigned)tiles), block(512);
      if (nk <= 4 && tiles > 512) {
        if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 1, true>), grid, block, 0, stream, d, ab, bb);
        else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 1, true>), grid, block, 0, stream, d, ab, bb);
      } else if (tiles > 256) {
        if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 2, true>), grid, block, 0, stream, d, ab, bb);
        else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 2, true>), grid, block, 0, stream, d, ab, bb);
      } else {
        if (n128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 4, true>), grid, block, 0, stream, d, ab, bb);
        else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 4, true>), grid, block, 0, stream, d, ab, bb);
      }
      return true;
    }
  }
  static const int ns1_nk = [] { const char* e = getenv("GIC_TILE8_NS1_NK"); return e ? atoi(e) : 4; }();
  const int nk8 = cdiv(d.K, 64);
  if (EPI != EPI_HIGHWAY && nk8 <= ns1_nk && (d.N >= 128 ? big_tiles : (long)cdiv(d.M, 128) * cdiv(d.N, 64)) > 512) {
    if (d.N >= 128) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 1>), dim3((unsigned)big_tiles), dim3(512), 0, stream, d, ab, bb);
    else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 1>), dim3((unsigned)(cdiv(d.M, 128) * cdiv(d.N, 64))), dim3(512), 0, stream, d, ab, bb);
    return true;
  }
  if (d.N >= 128 && big_tiles >= big_min) {
    if (big_tiles > ns2_tiles) hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 2>), dim3((unsigned)big_tiles), dim3(512), 0, stream, d, ab, bb);
    else hipLaunchKernelGGL((tile8_kernel<TO, 128, EPI, CONV, 4>), dim3((unsigned)big_tiles), dim3(512), 0, stream, d, ab, bb);
  } else {
    const long tiles = (long)cdiv(d.M, 128) * cdiv(d.N, 64);
    if (tiles > ns2_tiles) hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 2>), dim3((unsigned)tiles), dim3(512), 0, stream, d, ab, bb);
    else hipLaunchKernelGGL((tile8_kernel<TO, 64, EPI, CONV, 4>), dim3((unsigned)tiles), dim3(512), 0, stream, d, ab, bb);
  }
  return true;
}


__global__ void zero2d_kernel(float* __restrict__ C, long ldc, int M, int N) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    C[(i / N) * ldc + (i % N)] = 0.f;
}

template <typename TI, typename TO, bool AKC, bool BKC, int BM, int BN, bool VEC, int EPI, bool CONV = false>
int launch(const GemmDesc& d, hipStream_t stream) {
  constexpr int BK = 128 / (int)sizeof(TI);
  const int tiles = cdiv(d.M, BM) * cdiv(d.N, BN);
  const int nk = cdiv(d.K, BK);
  // split-K only where the tile grid leaves most of the 256 CUs idle and K is deep enough to share
  int splits = 1;
  // (bf16 compute mode only: the f32 parity mode stays bit-reproducible run to run, atomics reorder the f32 sum)
  if (EPI == EPI_PLAIN && sizeof(TI) == 2 && sizeof(TO) == 4) {
    if (tiles < 24 && nk >= 8) {                 // skinny recurrent / head products: fill the chip
      splits = 256 / tiles;
      if (splits > nk / 2) splits = nk / 2;
      if (splits > 16) splits = 16;
    } else if (tiles <= 192 && nk >= 32) {       // deep-K gradients over few tiles (K = B*L or V): ~2 blocks per CU
      splits = 512 / tiles;
      if (splits > nk / 8) splits = nk / 8;
      if (splits > 16) splits = 16;
    } else if (!AKC && !BKC && tiles <= 192 && nk >= 16) {   // weight-gradient form (both operands row-major over K = B*L rows) over few tiles:
      splits = 512 / tiles;                      // the D embedding gradient, 157 tiles x K = 1280
      if (splits > nk / 4) splits = nk / 4;
    } else if (tiles <= 320 && nk >= 64) {       // about one 4-wave block per CU over a very deep K (the highway weight gradient: 225 tiles,
      splits = 768 / tiles;                      // K = 2B*R = 8192): ~3 blocks per CU, 87 -> 47 us (tools/gemm_hw_bench.py)
    }
    if (splits < 1) splits = 1;
  }
  int per = cdiv(nk, splits);
  splits = cdiv(nk, per);
  if (splits > 1 && !d.accumulate && !d.c_zeroed) {
    const long total = (long)d.M * d.N;
    const int g = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(zero2d_kernel, dim3(g), dim3(256), 0, stream, (float*)d.C, d.ldc, d.M, d.N);
  }
  // LDS-DMA ring (PIPE) vs register staging: the ring hides load latency when a CU holds ONE block (grid <= ~2 blocks
  // per CU) and K is deep; with many blocks per CU the register-staged kernel wins (2 co-resident blocks, 72 KB LDS each)
  // and for K of one or two tiles the ring's prologue is pure overhead.  Measured on MI355X (tools/gemm_bench.py).
  constexpr bool CAN_PIPE = AKC && BKC && VEC;
  static const bool no_pipe = getenv("GIC_GEMM_NO_PIPE") != nullptr;
  const bool pipe = CAN_PIPE && per >= 6 && (long)tiles * splits <= 2 * 256 + 8 && !no_pipe;
  if constexpr (CAN_PIPE) {
    if (pipe) {
      hipLaunchKernelGGL((gemm_kernel<TI, TO, AKC, BKC, BM, BN, VEC, EPI, CONV, true>), dim3(tiles, splits), dim3(256), 0, stream, d, per);
      GIC_CHECK_LAUNCH("gemm");
      return GIC_OK;
    }
  }
  hipLaunchKernelGGL((gemm_kernel<TI, TO, AKC, BKC, BM, BN, VEC, EPI, CONV, false>), dim3(tiles, splits), dim3(256), 0, stream, d, per);
  GIC_CHECK_LAUNCH("gemm");
  return GIC_OK;
}

// 128x128 tiles only when there are enough of them to co-schedule two blocks per CU (latency hiding by TLP);
// GIC_GEMM_BIG_MIN overrides the threshold for tuning runs.
int big_tile_min() {
  static const int v = [] { const char* e = getenv("GIC_GEMM_BIG_MIN"); return e ? atoi(e) : 192; }();
  return v;
}

template <typename TI, typename TO, bool AKC, bool BKC, int EPI>
int pick_tile(const GemmDesc& d, bool vec, hipStream_t stream) {
  // 128x128 tiles when they still give >= ~1 block per CU, else 64x64.
  const long big_tiles = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
  const bool big = big_tiles >= big_tile_min() && d.M >= 128 && d.N >= 128;
  if (big) {
    return vec ? launch<TI, TO, AKC, BKC, 128, 128, true, EPI>(d, stream)
               : launch<TI, TO, AKC, BKC, 128, 128, false, EPI>(d, stream);
  }
  return vec ? launch<TI, TO, AKC, BKC, 64, 64, true, EPI>(d, stream)
             : launch<TI, TO, AKC, BKC, 64, 64, false, EPI>(d, stream);
}

template <typename TI, typename TO, int EPI>
int pick_layout(const GemmDesc& d, bool vec, hipStream_t stream) {
  if constexpr (sizeof(TI) == 2) {
    if (d.a_kc && d.b_kc && vec && try_tile8<TO, EPI, false>(d, stream)) { GIC_CHECK_LAUNCH("gemm tile8"); return GIC_OK; }
  }
  if (d.a_kc && d.b_kc) return pick_tile<TI, TO, true, true, EPI>(d, vec, stream);
  if constexpr (EPI == EPI_PLAIN) {
    if (d.a_kc && !d.b_kc) return pick_tile<TI, TO, true, false, EPI>(d, vec, stream);
    if (!d.a_kc && !d.b_kc) return pick_tile<TI, TO, false, false, EPI>(d, vec, stream);
  }
  set_last_error("gemm: unsupported layout a_kc=%d b_kc=%d epi=%d", d.a_kc, d.b_kc, d.epi);
  return GIC_ERR_UNSUPPORTED;
}

template <typename TI, typename TO, int EPI>
int pick_conv(const GemmDesc& d, hipStream_t stream) {
  if constexpr (sizeof(TI) == 2 && sizeof(TO) == 2 && EPI == EPI_BNSTATS) {
    // 3x3 / stride 1: the input patch stays in LDS for all nine taps (conv3x3.hip).  BatchNorm on load of any other window than
    // 1x1 exists there only (tile8 would re-normalise the tile once per tap: slower than the separate pass it replaces).
    // the stem (7 x 8 window over the zero-bordered NHWC4 image): input rows rolling through an LDS ring (conv_stem.hip)
    if (try_conv_stem(d, stream)) { GIC_CHECK_LAUNCH("conv stem"); return GIC_OK; }
    if (try_conv3x3_patch(d, stream)) { GIC_CHECK_LAUNCH("conv3x3 patch"); return GIC_OK; }
    // shallow 1x1 layers over many rows: persistent workgroups, resident weights, A tiles streamed across row tiles (conv1x1_stream.hip)
    if (try_conv1x1_stream(d, stream)) { GIC_CHECK_LAUNCH("conv1x1 stream"); return GIC_OK; }
    if (d.stats_only) return GIC_ERR_UNSUPPORTED;               // (no message: callers probe)
    // K = 256 | 512 of a normalised input into many output channels: the pixels in registers, weight tiles streamed (conv1x1_pix.hip)
    if (try_conv1x1_pix(d, stream)) { GIC_CHECK_LAUNCH("conv1x1 pix"); return GIC_OK; }
    // K = 256 into many output channels: the A panel of a row tile loaded / normalised once for all its channel tiles (conv1x1_panel.hip)
    if (try_conv1x1_panel(d, stream)) { GIC_CHECK_LAUNCH("conv1x1 panel"); return GIC_OK; }
    if (d.in_stats && d.cKH * d.cKW > 1) return GIC_ERR_UNSUPPORTED;
  }
  if constexpr (sizeof(TI) == 2) {
    if (try_tile8<TO, EPI, true>(d, stream)) { GIC_CHECK_LAUNCH("conv tile8"); return GIC_OK; }
  }
  if (d.in_stats) return GIC_ERR_UNSUPPORTED;                 // the A-side BatchNorm exists in tile8 only (no message: callers probe)
  const long big_tiles = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
  if (big_tiles >= big_tile_min() && d.N >= 128) return launch<TI, TO, true, true, 128, 128, true, EPI, true>(d, stream);
  return launch<TI, TO, true, true, 64, 64, true, EPI, true>(d, stream);
}

template <typename TI, typename TO>
int pick_epi(const GemmDesc& d, bool vec, hipStream_t stream) {
  if (d.conv) {
    if (!vec || !d.a_kc || !d.b_kc) { set_last_error("gemm: convolution needs 16-B aligned NHWC / KRSC operands"); return GIC_ERR_UNSUPPORTED; }
    if constexpr (sizeof(TI) == sizeof(TO)) {
      if (d.epi == EPI_BNSTATS) return pick_conv<TI, TO, EPI_BNSTATS>(d, stream);
      if (d.epi == EPI_PLAIN) return pick_conv<TI, TO, EPI_PLAIN>(d, stream);
    }
    set_last_error("gemm: unsupported convolution epilogue / dtype");
    return GIC_ERR_UNSUPPORTED;
  }
  if (d.epi == EPI_PLAIN) return pick_layout<TI, TO, EPI_PLAIN>(d, vec, stream);
  if (d.epi == EPI_HIGHWAY) return pick_layout<TI, TO, EPI_HIGHWAY>(d, vec, stream);
  set_last_error("gemm: unknown epilogue %d", d.epi);
  return GIC_ERR_UNSUPPORTED;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

int gemm_gumbelmax(const GemmDesc& d0, hipStream_t stream) {
  GemmDesc d = d0;
  static const bool off = getenv("GIC_NO_TILE8") != nullptr || getenv("GIC_NO_FUSED_GUMBELMAX") != nullptr;
  GIC_CHECK_ARG(d.A && d.B && d.gm_rowkey && d.gm_bias, "gemm_gumbelmax: null operand");
  if (off || d.in_dtype != DT_BF16 || !d.a_kc || !d.b_kc || d.M < 128 || d.M % 4 || d.K % 8 || d.lda % 8 || d.ldb % 8 || !aligned16(d.A) ||
      !aligned16(d.B) || !aligned16(d.gm_bias) || (d.gm_u && (!aligned16(d.gm_u) || d.gm_ldu % 4)) || !(d.gm_temperature > 0.f))
    return GIC_ERR_UNSUPPORTED;
  const long a_elems = (long)(d.M - 1) * d.lda + d.K, b_elems = (long)(d.N - 1) * d.ldb + d.K;
  const long tiles = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
  if (a_elems * 2 >= (1l << 31) || b_elems * 2 >= (1l << 31) || d.N < 128 || tiles < 160) return GIC_ERR_UNSUPPORTED;
  d.epi = EPI_GUMBELMAX;
  d.n_fast = xcd_share_a(2l * d.M * d.K, 2l * d.N * d.K, cdiv(d.N, 128));
  const unsigned ab = (unsigned)(a_elems * 2), bb = (unsigned)(b_elems * 2);
  if (tiles > 256) hipLaunchKernelGGL((tile8_kernel<float, 128, EPI_GUMBELMAX, false, 2>), dim3((unsigned)tiles), dim3(512), 0, stream, d, ab, bb);
  else hipLaunchKernelGGL((tile8_kernel<float, 128, EPI_GUMBELMAX, false, 4>), dim3((unsigned)tiles), dim3(512), 0, stream, d, ab, bb);
  GIC_CHECK_LAUNCH("gemm gumbelmax");
  return GIC_OK;
}

int gemm(const GemmDesc& d0, hipStream_t stream) {
  GemmDesc d = d0;
#ifdef GIC_STAMPS
  { const char* e = getenv("GIC_GEMM_DBG"); if (e) d.dbg = atoi(e); }
#endif
  GIC_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand");
  GIC_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0, "gemm: negative dim");
  if (d.M == 0 || d.N == 0) return GIC_OK;
  {  // GIC_GEMM_LOG=1: one line per product on stderr (tools: which shapes a step issues)
    static const bool log = getenv("GIC_GEMM_LOG") != nullptr;
    if (log && !d.conv)
      fprintf(stderr, "[gemm] M=%d N=%d K=%d a_kc=%d b_kc=%d in=%d out=%d epi=%d acc=%d\n", d.M, d.N, d.K, d.a_kc, d.b_kc, d.in_dtype, d.out_dtype, d.epi,
              d.accumulate);
  }
  if (d.epi == EPI_HIGHWAY) GIC_CHECK_ARG(d.X, "gemm: highway epilogue needs X");      // Hpre may be null (forward only)
  const int sz = dtype_size(d.in_dtype);
  const int ve = 16 / sz;
  bool vec = aligned16(d.A) && aligned16(d.B) && (d.lda % ve == 0) && (d.ldb % ve == 0);
  if (d.conv) {
    // a 16-B chunk holds `ve` consecutive k = channels of one tap, or (pre-padded input, pad == 0) whole adjacent taps of one row
    const bool chunk_ok = (d.cCin % ve == 0) || (d.cPad == 0 && ve % d.cCin == 0 && (d.cKW * d.cCin) % ve == 0);
    GIC_CHECK_ARG(chunk_ok && d.K == d.cKH * d.cKW * d.cCin && d.K % ve == 0, "gemm: conv needs Cin %% %d == 0 (or a pre-padded NHWC4 stem) and K = KH*KW*Cin", ve);
    GIC_CHECK_ARG(d.epi != EPI_BNSTATS || d.stats, "gemm: EPI_BNSTATS needs a stats buffer");
    vec = aligned16(d.A) && aligned16(d.B) && (d.ldb % ve == 0);
  }
  // k-contiguous operands: K must be a whole number of 16-B chunks (callers zero-pad K).  m/n-contiguous
  // operands: a tail chunk reads into the row's padding (ld % ve == 0 >= M) and only feeds rows that are
  // never stored, so no condition on M / N.
  if (d.a_kc || d.b_kc) vec = vec && (d.K % ve == 0);
  if (d.in_dtype == DT_F32 && d.out_dtype == DT_F32) return pick_epi<float, float>(d, vec, stream);
  if (d.in_dtype == DT_BF16 && d.out_dtype == DT_F32) return pick_epi<bf16_t, float>(d, vec, stream);
  if (d.in_dtype == DT_BF16 && d.out_dtype == DT_BF16) return pick_epi<bf16_t, bf16_t>(d, vec, stream);
  set_last_error("gemm: unsupported dtypes in=%d out=%d", d.in_dtype, d.out_dtype);
  return GIC_ERR_UNSUPPORTED;
}

}  // namespace gic
#ifdef GIC_STAMPS
extern "C" int gic_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gic::g_stamp), sizeof(unsigned long long) * 32);
}
#endif
