// 1x1 convolutions with K = 256 input channels and many output channels (conv3 of the 14x14 bottlenecks of the ResNet trunk: 256 ->
// 1024, reference src/generator.py:12-14) with the A PANEL of a 128-row tile RESIDENT in LDS: loaded (and, with BatchNorm + ReLU on
// load, normalised) once, then multiplied with the output-channel tiles of the workgroup's group one after the other while their
// weight tiles stream through a two-stage LDS-DMA ring.
//
// tile8 runs such a layer as 784 independent 128x128 tiles: every one of the eight tiles of a row block fetches the same 64 KB of
// A, rewrites it with the same coefficients and pays its own coefficient table, DMA round trips (ring-less: one per K tile) and
// drain (measured 27 us per launch against an HBM floor of 4 us, 11.5 us per workgroup of which 7 us are the serial K tiles).
#include <stdlib.h>

#include "conv1x1_panel.h"
#include "bn_fold.h"

namespace gic {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// everything the kernel reads from its arguments, compact (one batch of scalar loads at the top)
struct PanelDesc {
  const void* A; const void* B; void* C; float* stats;
  const float* in_stats; const float* in_gamma; const float* in_beta;
  int M, N, lda, ldb, ldc, stats_nrep, in_nrep;
  float in_inv_count;
  int share_a;
  int tiles_m, tiles_n, groups, per_group;   // row tiles, 64-wide output-channel tiles, groups of them, tiles per group
  unsigned a_bytes, b_bytes;
};

// KT = K / 64 (4); ABN: BatchNorm + ReLU of the input on load.  8 waves as 4 (M) x 2 (N): a wave owns 32 x 32 of a 128 x 64 output tile.
template <int KT, bool ABN>
__global__ __launch_bounds__(512) void conv1x1_panel_kernel(const PanelDesc d) {
  constexpr int BM = 128, BN = 64, NT = 512, K = 64 * KT;
  constexpr int TM = 2, TN = 2;
  constexpr int CA = KT * 2;                                            // 16-byte pieces of the A panel per thread
  constexpr int CW = KT;                                                // ... of a weight tile (64 rows x 128 B per K tile)
  constexpr int CS = BM * BN * 2 / 16 / NT;                             // 16-byte stores of a C tile per thread (2)
  constexpr int A_BYTES = KT * BM * 128, W_BYTES = KT * BN * 128;
  constexpr int SC = BN * 2 + 16, C_BYTES = BM * SC;
  constexpr int W0 = A_BYTES, C0 = W0 + 2 * W_BYTES, ST0 = C0 + C_BYTES, COEF0 = ST0 + 8 * (BN / 2) * 2 * 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int M = d.M, N = d.N;
  // workgroup -> (row tile, group of output-channel tiles): the groups of one row tile are neighbours on one XCD (they share the A panel
  // in its L2: xcd_run)
  const int bid = d.share_a ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
  const int grp = bid % d.groups, tile_m = bid / d.groups;
  const int bm0 = tile_m * BM;
  const int nt0 = grp * d.per_group;
  const int nt1 = min(nt0 + d.per_group, d.tiles_n);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)d.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, (int)d.b_bytes, 0x00020000);
  const int wbase = (tid & ~63) * 16;
  const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;                    // this thread's 8 channels inside every 64-channel K tile

  // ---- prologue: the A panel (piece i: K tile i / 2, rows (tid >> 3) + 64 (i & 1)), the first weight tile
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int row = bm0 + (tid >> 3) + 64 * (i & 1);
    const unsigned voff = row < M ? (unsigned)(row * d.lda + (i >> 1) * 64 + kc) * 2u : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(smem + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, GIC_TRUNK_NT);
  }
  auto issue_w = [&](const int nt, const int st) {                      // weight tile nt: piece i = K tile i, rows tid >> 3
    const int n = nt * BN + (tid >> 3);
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      const unsigned voff = (nt < nt1 && n < N) ? (unsigned)(n * d.ldb + i * 64 + kc) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(smem + W0 + st * W_BYTES + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, 0);
    }
  };
  issue_w(nt0, 0);

  if constexpr (ABN) {
    float* coef = (float*)(smem + COEF0);
    for (int c = tid; c < K; c += NT) {
      const float gam = d.in_gamma[c], bet = d.in_beta[c];
      float s1, s2;
      fold_replicas(d.in_stats, d.in_nrep, K, c, s1, s2);
      const float mean = s1 * d.in_inv_count;
      const float var = fmaxf(s2 * d.in_inv_count - mean * mean, 0.f);
      const float sc = gam * rsqrtf(var + 1e-5f);                        // kBnEps of encoder.hip (nn.BatchNorm2d default)
      coef[2 * c] = sc;
      coef[2 * c + 1] = bet - mean * sc;
    }
    __syncthreads();
    wait_vm<CW>();                                                       // the panel has landed (the weight tile may be in flight)
    // a thread's pieces of K tile kt all hold channels kt*64 + kc .. + 7; rows past M were zero-filled and stay zero
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      if (bm0 + (tid >> 3) + 64 * (i & 1) >= M) continue;
      const float4* cp = (const float4*)(coef + 2 * ((i >> 1) * 64 + kc));
      const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
      const float scl[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
      const float sft[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
      bf16x8* p = (bf16x8*)(smem + (tid + NT * i) * 16);
      bf16x8 v = *p;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)fmaxf((float)v[e] * scl[e] + sft[e], 0.f);
      *p = v;
    }
  }

  bf16_t* __restrict__ C = (bf16_t*)d.C;
  unsigned char* sC = smem + C0;
  float* sStat = (float*)(smem + ST0);                                   // [8 waves][BN/2][2]
  int st = 0;
  for (int nt = nt0; nt < nt1; ++nt) {
    const int bn0 = nt * BN;
    // weight tile nt: behind it came only the previous tile's stores (wave 0: + its two atomics; waiting for all but CS is the
    // stricter count there).  The first tile waits for everything (panel included).
    // A partial row tile skips stores in some waves (their counts differ): it waits for everything.
    if (nt == nt0 || bm0 + BM > M) wait_vm<0>(); else wait_vm<CS>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // weights (and, first tile, the normalised panel) published; the other stage and the C tile are free
    issue_w(nt + 1, st ^ 1);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* sW = smem + W0 + st * W_BYTES;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = wr * 32 + t * 16 + lr;
          fa[t] = *(const bf16x8*)(smem + kt * (BM * 128) + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int row = wc * (BN / 2) + t * 16 + lr;
          fb[t] = *(const bf16x8*)(sW + kt * (BN * 128) + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    // C tile through LDS, BatchNorm column sums folded across the workgroup (as tile8)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wc * (BN / 2) + j * 16 + lr;
      float st_s = 0.f, st_q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ml = wr * 32 + i * 16 + lg * 4 + r;
          const float v = acc[i][j][r];
          *(bf16_t*)(sC + ml * SC + nl * 2) = (bf16_t)v;
          if (bm0 + ml < M) { st_s += v; st_q += v * v; }
        }
      }
      st_s += __shfl_xor(st_s, 16, 64); st_q += __shfl_xor(st_q, 16, 64);
      st_s += __shfl_xor(st_s, 32, 64); st_q += __shfl_xor(st_q, 32, 64);
      if (lg == 0) { sStat[(w * (BN / 2) + j * 16 + lr) * 2] = st_s; sStat[(w * (BN / 2) + j * 16 + lr) * 2 + 1] = st_q; }
    }
    __syncthreads();
    if (tid < BN) {                                                      // column tid (all of wave 0): waves (wr = 0..3, wc)
      const int cwc = tid / (BN / 2), cl = tid % (BN / 2), n = bn0 + tid;
      float s0 = 0.f, q0 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0 += sStat[((r * 2 + cwc) * (BN / 2) + cl) * 2];
        q0 += sStat[((r * 2 + cwc) * (BN / 2) + cl) * 2 + 1];
      }
      if (n < N) {
        float* stp = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * N;
        atomicAdd(&stp[n], s0);
        atomicAdd(&stp[N + n], q0);
      }
    }
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int i = 0; i < CS; ++i) {
      const int c = tid + NT * i;
      const int ml = c / CPR, cc = c % CPR;
      const int m = bm0 + ml, n = bn0 + cc * 8;
      const u32x4 v = *(const u32x4*)(sC + ml * SC + cc * 16);
      if (m < M && n < N) *(u32x4*)(C + (long)m * d.ldc + n) = v;
    }
    st ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the zero-fill DMA of the tile past the group)
}

template <int KT, bool ABN>
bool launch_panel(const PanelDesc& pd, hipStream_t stream) {
  constexpr size_t lds = (size_t)KT * 128 * 128 + 2 * (size_t)KT * 64 * 128 + 128 * (64 * 2 + 16) + 8 * 32 * 2 * 4 + (ABN ? 64 * KT * 8 : 0);
  static LdsGrant granted;
  if (!grant_lds(conv1x1_panel_kernel<KT, ABN>, lds, granted)) return false;
  hipLaunchKernelGGL((conv1x1_panel_kernel<KT, ABN>), dim3((unsigned)(pd.tiles_m * pd.groups)), dim3(512), lds, stream, pd);
  return true;
}

}  // namespace

bool try_conv1x1_panel(const GemmDesc& d, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_CONV1X1_PANEL") != nullptr;
  if (off || !d.conv || d.epi != EPI_BNSTATS || !d.stats || d.res) return false;
  if (d.in_dtype != DT_BF16 || d.out_dtype != DT_BF16) return false;
  if (d.cKH != 1 || d.cKW != 1 || d.cStride != 1 || d.cPad != 0) return false;
  if (d.K != 256 || d.cCin != d.K || d.lda != d.K || d.N < 512 || d.N % 64) return false;      // the panel pays where it serves many tiles
  if (d.ldc % 8 || d.ldb % 8 || (((uintptr_t)d.C) & 15) || (((uintptr_t)d.A) & 15) || (((uintptr_t)d.B) & 15)) return false;
  if (d.bias || d.alpha != 1.f || d.accumulate) return false;
  const bool abn = d.in_stats != nullptr;
  if (abn && (!d.in_gamma || !d.in_beta || d.in_inv_count <= 0.f || d.in_nrep < 1)) return false;
  const long a_elems = (long)d.M * d.K, b_elems = (long)(d.N - 1) * d.ldb + d.K;
  if (a_elems * 2 >= (1l << 31) || b_elems * 2 >= (1l << 31)) return false;
  PanelDesc pd;
  pd.tiles_m = cdiv(d.M, 128);
  pd.tiles_n = d.N / 64;
  // about one workgroup per CU: split the output-channel tiles of a row tile over as many groups as that leaves room for
  int groups = 256 / pd.tiles_m;
  if (groups < 1) groups = 1;
  if (groups > pd.tiles_n) groups = pd.tiles_n;
  pd.per_group = cdiv(pd.tiles_n, groups);
  pd.groups = cdiv(pd.tiles_n, pd.per_group);
  pd.share_a = xcd_share_a(2l * d.M * d.K, 2l * d.N * d.K, pd.groups);
  pd.A = d.A; pd.B = d.B; pd.C = d.C; pd.stats = d.stats;
  pd.in_stats = d.in_stats; pd.in_gamma = d.in_gamma; pd.in_beta = d.in_beta;
  pd.M = d.M; pd.N = d.N; pd.lda = (int)d.lda; pd.ldb = (int)d.ldb; pd.ldc = (int)d.ldc;
  pd.stats_nrep = d.stats_nrep < 1 ? 1 : d.stats_nrep; pd.in_nrep = d.in_nrep; pd.in_inv_count = d.in_inv_count;
  pd.a_bytes = (unsigned)(a_elems * 2); pd.b_bytes = (unsigned)(b_elems * 2);
  return abn ? launch_panel<4, true>(pd, stream) : launch_panel<4, false>(pd, stream);
}

}  // namespace gic
