// Shallow-K 1x1 convolutions of the ResNet trunk (torchvision conv1x1 of the bottleneck blocks, reference src/generator.py:12-14)
// as a STREAMING kernel: K = 64 or 128 input channels, a large row count, bf16, BatchNorm column sums, optional BatchNorm + ReLU of
// the input on load.
//
// With K this shallow a 128x128 output tile is 0.1 us of MFMA work behind 64-96 KB of traffic: the layer is bound by HBM
// (16 us for the widest one at 8 TB/s), and tile8 runs it at a third of that -- one workgroup per tile, each a serial chain
// (coefficient table, DMA round trip, BatchNorm rewrite, MFMAs, C tile through LDS, column sums, atomics, stores) with two workgroups
// per CU to overlap (measured with tools/conv_stamps.py: 7.5 us per workgroup, 3 of them epilogue, 1.5 prologue).  Here a
// workgroup is persistent: it keeps its 128 (or 64) output channels' weights in LDS, walks row tiles with a stride of the grid,
// streams their A tiles through an LDS-DMA ring two or three tiles ahead, accumulates the BatchNorm column sums in registers
// across all its tiles (one atomic per column and workgroup at the end) and pays the prologue once.
#include <stdlib.h>

#include "conv1x1_stream.h"
#include "bn_fold.h"

namespace gic {
namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// everything the kernel reads from its arguments, compact (one batch of scalar loads at the top)
struct StreamDesc {
  const void* A; const void* B; void* C; float* stats;
  const float* in_stats; const float* in_gamma; const float* in_beta;
  int M, N, lda, ldb, ldc, stats_nrep, in_nrep;
  float in_inv_count;
  int share_a;
  int stats_only;                      // column sums only: no C stores
  int tiles_m, tiles_n, groups;        // row tiles, output-channel tiles, workgroups per output-channel tile (grid = groups * tiles_n)
  unsigned a_bytes, b_bytes;
};

// BN output channels per workgroup; KT = K / 64; ABN: BatchNorm + ReLU of the input on load; STATS: the column sums ONLY (the first pass of
// conv_b2b.hip's pair: no C tile in LDS, no stores -- which leaves room for 256 output channels per workgroup, so that a row tile is
// fetched and normalised once for all of them)
template <int BN, int KT, bool ABN, bool STATS = false>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(const StreamDesc d) {
  constexpr int BM = 128, NT = 512, K = 64 * KT;
  constexpr int NS = KT == 1 ? 4 : 2;                                   // ring stages of A tiles (NS - 1 tiles in flight)
  constexpr int TM = 2, TN = BN / 32;
  constexpr int CA = KT * 2;                                            // 16-byte pieces of an A tile per thread (128 rows x 128 B per K tile)
  constexpr int CW = KT * BN / 64;                                      // ... of the weight tile
  constexpr int CS = BM * BN * 2 / 16 / NT;                             // 16-byte stores of a C tile per thread
  constexpr int A_BYTES = KT * BM * 128, W_BYTES = KT * BN * 128;
  constexpr int SC = BN * 2 + 16, C_BYTES = BM * SC;
  constexpr bool DBUF = KT == 1;                                        // two C tiles: one barrier per row tile, the stores of a tile run under the next tile's MFMAs
  constexpr int W0 = NS * A_BYTES, C0 = W0 + W_BYTES, ST0 = C0 + (STATS ? 0 : (DBUF ? 2 : 1) * C_BYTES), COEF0 = ST0 + 4 * BN * 2 * 4;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int M = d.M, N = d.N;
  // workgroup -> (output-channel tile, first row tile): the channel tiles of one row-tile sequence are neighbours on one XCD and walk
  // their row tiles in step, so each A tile leaves HBM once (share_a; the weights, <= 128 KB, sit in every L2 anyway)
  const int bid = d.share_a ? xcd_run(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tile_n = bid % d.tiles_n;
  const int grp = bid / d.tiles_n;
  const int bn0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)d.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, (int)d.b_bytes, 0x00020000);
  const int wbase = (tid & ~63) * 16;
  // piece i of a tile: K tile i / 2 (A) , rows (tid >> 3) + 64 (i & 1), physical slot tid & 7 = logical 16-byte chunk ^ ((row >> 1) & 7)
  const int kc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;                    // this thread's 8 channels inside every 64-channel K tile
  auto issue_a = [&](const int tile, const int st) {
    const int bm0 = tile * BM;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int row = bm0 + (tid >> 3) + 64 * (i & 1);
      const unsigned voff = (tile < d.tiles_m && row < M) ? (unsigned)(row * d.lda + (i >> 1) * 64 + kc) * 2u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_ptr)(smem + st * A_BYTES + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, GIC_TRUNK_NT);
    }
  };
  // ---- prologue: the weights of this workgroup's output channels (resident), the first NS - 1 row tiles
  constexpr int WR = BN / 64;                                           // 64-row groups of the weight tile
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    const int n = bn0 + (tid >> 3) + 64 * (i % WR);
    const unsigned voff = n < N ? (unsigned)(n * d.ldb + (i / WR) * 64 + kc) * 2u : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_ptr)(smem + W0 + i * (NT * 16) + wbase), 16, (int)voff, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_a(grp + s * d.groups, s);

  // ---- A-side BatchNorm + ReLU: this thread's pieces always hold the same 8 channels of each K tile -> coefficients in registers
  float scl[KT][8], sft[KT][8];
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
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const float4* cp = (const float4*)(coef + 2 * (kt * 64 + kc));
      const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
      scl[kt][0] = c0.x; scl[kt][1] = c0.z; scl[kt][2] = c1.x; scl[kt][3] = c1.z; scl[kt][4] = c2.x; scl[kt][5] = c2.z; scl[kt][6] = c3.x; scl[kt][7] = c3.z;
      sft[kt][0] = c0.y; sft[kt][1] = c0.w; sft[kt][2] = c1.y; sft[kt][3] = c1.w; sft[kt][4] = c2.y; sft[kt][5] = c2.w; sft[kt][6] = c3.y; sft[kt][7] = c3.w;
    }
  }

  float st_s[TN], st_q[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) st_s[j] = st_q[j] = 0.f;
  bf16_t* __restrict__ C = (bf16_t*)d.C;
  unsigned char* sC = smem + C0;

  // ---- row tiles grp, grp + groups, ...: ring stage of the j-th = j % NS
  constexpr int CPR = BN / 8;                                            // 16-byte chunks per C tile row
  auto store_tile = [&](const int bm0, const unsigned char* buf) {
    if constexpr (STATS) return;                                         // the statistics-only pass writes (and stages) nothing
#pragma unroll
    for (int i = 0; i < CS; ++i) {
      const int c = tid + NT * i;
      const int ml = c / CPR, cc = c % CPR;
      const int m = bm0 + ml, n = bn0 + cc * 8;
      const u32x4 v = *(const u32x4*)(buf + ml * SC + cc * 16);
      if (m < M && n < N) *(u32x4*)(C + (long)m * d.ldc + n) = v;      // (a wave skips a store only in its last, partial row tile)
    }
  };
  int st = 0, jj = 0, prev_bm0 = 0;
  for (int tile = grp; tile < d.tiles_m; tile += d.groups) {
    const int bm0 = tile * BM;
    // A tile j has landed once all but the DMAs and stores issued after it are done.  Per iteration a thread issues the DMA of the tile
    // NS - 1 ahead (CA pieces), then the stores of a C tile (CS; double-buffered C: of the PREVIOUS tile, so iteration 0 has none).
    if (STATS) {               // no stores in the queue: only the DMAs issued behind tile j may be outstanding
      if constexpr (DBUF) wait_vm<2 * CA>(); else wait_vm<0>();
    } else if constexpr (DBUF) {      // NS = 4: behind tile j came 2 tiles' DMAs and the stores of iterations max(1, j - 3) .. j - 1
      if (jj <= 1) wait_vm<2 * CA>();
      else if (jj == 2) wait_vm<2 * CA + CS>();
      else if (jj == 3) wait_vm<2 * CA + 2 * CS>();
      else wait_vm<2 * CA + 3 * CS>();
    } else {                   // NS = 2: tile j was issued in iteration j - 1 ahead of that iteration's stores
      if (jj == 0) wait_vm<0>(); else wait_vm<CS>();
    }
    if constexpr (ABN) {
      // rows past M were zero-filled and must stay zero (they feed rows that are never stored; keeps the sums exact)
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        if (bm0 + (tid >> 3) + 64 * (i & 1) >= M) continue;
        bf16x8* p = (bf16x8*)(smem + st * A_BYTES + (tid + NT * i) * 16);
        bf16x8 v = *p;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)fmaxf((float)v[e] * scl[i >> 1][e] + sft[i >> 1][e], 0.f);
        *p = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // A tile published; the previous C tile is complete; every wave is done with ring stage st - 1
    // the tile NS - 1 ahead goes into the stage that the previous tile occupied
    issue_a(tile + (NS - 1) * d.groups, st == 0 ? NS - 1 : st - 1);
    unsigned char* sCj = sC + (DBUF ? (jj & 1) * C_BYTES : 0);
    if constexpr (DBUF && !STATS) {
      if (jj > 0) store_tile(prev_bm0, sC + ((jj - 1) & 1) * C_BYTES);   // runs under this tile's MFMAs
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* sA = smem + st * A_BYTES;
    const unsigned char* sW = smem + W0;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int row = wr * 32 + t * 16 + lr;
          fa[t] = *(const bf16x8*)(sA + kt * (BM * 128) + row * 128 + (((ks * 4 + lg) ^ ((row >> 1) & 7)) << 4));
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
    // C tile -> LDS, column sums into the registers that live across tiles
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wc * (BN / 2) + j * 16 + lr;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ml = wr * 32 + i * 16 + lg * 4 + r;
          const float v = acc[i][j][r];
          if constexpr (!STATS) *(bf16_t*)(sCj + ml * SC + nl * 2) = (bf16_t)v;
          if (bm0 + ml < M) { st_s[j] += v; st_q[j] += v * v; }
        }
      }
    }
    if constexpr (!DBUF && !STATS) {
      __syncthreads();
      store_tile(bm0, sCj);
    }
    prev_bm0 = bm0;
    ++jj;
    st = st == NS - 1 ? 0 : st + 1;
  }
  if constexpr (DBUF && !STATS) {
    __syncthreads();
    if (jj > 0) store_tile(prev_bm0, sC + ((jj - 1) & 1) * C_BYTES);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- BatchNorm column sums of all this workgroup's tiles: fold the 4 row groups of a wave, then the 4 row waves, one atomic per column
  float* sStat = (float*)(smem + ST0);                                   // [4 row waves][BN][2]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float s = st_s[j], q = st_q[j];
    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
    if (lg == 0) {
      const int nl = wc * (BN / 2) + j * 16 + lr;
      sStat[(wr * BN + nl) * 2] = s;
      sStat[(wr * BN + nl) * 2 + 1] = q;
    }
  }
  __syncthreads();
  if (tid < BN && bn0 + tid < N) {
    float s0 = 0.f, q0 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s0 += sStat[(r * BN + tid) * 2]; q0 += sStat[(r * BN + tid) * 2 + 1]; }
    float* stp = d.stats + (long)(blockIdx.x % d.stats_nrep) * 2 * N;
    atomicAdd(&stp[bn0 + tid], s0);
    atomicAdd(&stp[N + bn0 + tid], q0);
  }
}

template <int BN, int KT, bool ABN, bool STATS = false>
bool launch_stream(const StreamDesc& sd, hipStream_t stream) {
  constexpr int NS = KT == 1 ? 4 : 2;
  constexpr size_t lds = (size_t)NS * KT * 128 * 128 + (size_t)KT * BN * 128 + (STATS ? 0 : (KT == 1 ? 2 : 1) * 128 * (BN * 2 + 16)) + 4 * BN * 2 * 4 +
                         (ABN ? 64 * KT * 8 : 0);
  static_assert(lds <= 160 * 1024, "conv1x1_stream LDS budget");
  static LdsGrant granted;
  if (!grant_lds(conv1x1_stream_kernel<BN, KT, ABN, STATS>, lds, granted)) return false;
  hipLaunchKernelGGL((conv1x1_stream_kernel<BN, KT, ABN, STATS>), dim3((unsigned)(sd.groups * sd.tiles_n)), dim3(512), lds, stream, sd);
  return true;
}

}  // namespace

bool try_conv1x1_stream(const GemmDesc& d, hipStream_t stream) {
  static const bool off = getenv("GIC_NO_CONV1X1_STREAM") != nullptr;
  static const int min_tiles = [] { const char* e = getenv("GIC_STREAM_MIN_TILES"); return e ? atoi(e) : 1024; }();
  if (off || !d.conv || d.epi != EPI_BNSTATS || !d.stats || d.res) return false;
  if (d.in_dtype != DT_BF16 || d.out_dtype != DT_BF16) return false;
  if (d.cKH != 1 || d.cKW != 1 || d.cStride != 1 || d.cPad != 0) return false;
  if ((d.K != 64 && d.K != 128) || d.cCin != d.K || d.lda != d.K) return false;
  if (d.N % 8 || d.ldc % 8 || d.ldb % 8 || (((uintptr_t)d.C) & 15) || (((uintptr_t)d.A) & 15) || (((uintptr_t)d.B) & 15)) return false;
  if (d.bias || d.alpha != 1.f || d.accumulate) return false;
  const bool abn = d.in_stats != nullptr;
  if (abn && (!d.in_gamma || !d.in_beta || d.in_inv_count <= 0.f || d.in_nrep < 1)) return false;
  const long a_elems = (long)d.M * d.K, b_elems = (long)(d.N - 1) * d.ldb + d.K;
  if (a_elems * 2 >= (1l << 31) || b_elems * 2 >= (1l << 31) || (long)d.M * d.ldc * 2 >= (1l << 40)) return false;
  static const bool bn64 = getenv("GIC_STREAM_BN64") != nullptr;
  static const int wg_per_cu = [] { const char* e = getenv("GIC_STREAM_WG_PER_CU"); return e ? atoi(e) : 1; }();
  const bool n128 = d.N >= 128 && !bn64;
  // the statistics-only pass (input normalised on load, output channels a multiple of 256): 256 channels per workgroup
  const bool stats256 = d.stats_only && abn && d.N % 256 == 0;
  if (d.stats_only && !stats256) return false;
  StreamDesc sd;
  sd.tiles_m = cdiv(d.M, 128);
  sd.tiles_n = stats256 ? d.N / 256 : (n128 ? cdiv(d.N, 128) : cdiv(d.N, 64));
  // streaming pays where a workgroup walks several row tiles; small grids stay with tile8 (counted in 128-wide tiles for both forms)
  if ((long)sd.tiles_m * (stats256 ? d.N / 128 : sd.tiles_n) < min_tiles) return false;
  int groups = 256 * wg_per_cu / sd.tiles_n;           // one persistent workgroup per CU (the ring + the C tile fill most of its LDS)
  if (groups < 1) groups = 1;
  if (groups > sd.tiles_m) groups = sd.tiles_m;
  sd.groups = groups;
  sd.share_a = xcd_share_a(2l * d.M * d.K, 2l * d.N * d.K, sd.tiles_n);
  sd.A = d.A; sd.B = d.B; sd.C = d.C; sd.stats = d.stats; sd.stats_only = d.stats_only;
  sd.in_stats = d.in_stats; sd.in_gamma = d.in_gamma; sd.in_beta = d.in_beta;
  sd.M = d.M; sd.N = d.N; sd.lda = (int)d.lda; sd.ldb = (int)d.ldb; sd.ldc = (int)d.ldc;
  sd.stats_nrep = d.stats_nrep < 1 ? 1 : d.stats_nrep; sd.in_nrep = d.in_nrep; sd.in_inv_count = d.in_inv_count;
  sd.a_bytes = (unsigned)(a_elems * 2); sd.b_bytes = (unsigned)(b_elems * 2);
  const int kt = d.K / 64;
  if (stats256) return kt == 1 ? launch_stream<256, 1, true, true>(sd, stream) : launch_stream<256, 2, true, true>(sd, stream);
  if (abn) {
    if (n128) return kt == 1 ? launch_stream<128, 1, true>(sd, stream) : launch_stream<128, 2, true>(sd, stream);
    return kt == 1 ? launch_stream<64, 1, true>(sd, stream) : launch_stream<64, 2, true>(sd, stream);
  }
  if (n128) return kt == 1 ? launch_stream<128, 1, false>(sd, stream) : launch_stream<128, 2, false>(sd, stream);
  return kt == 1 ? launch_stream<64, 1, false>(sd, stream) : launch_stream<64, 2, false>(sd, stream);
}

}  // namespace gic
