// fp32 dense contraction engine for gfx950, two arithmetic modes (advmil_set_gemm_mode):
//   exact  : v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak)
//   bf16x3 : fp32 operands split hi + lo bf16 on the way into LDS, three v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate
//
// Workgroup = WR x WC waves (2x2: 256 threads; 4x2 / 2x4: 512 threads, bf16x3 only), each wave TM x TN MFMA 32x32 accumulators,
// k walked in chunks of 32. Operands are staged global -> registers -> LDS.
//
// exact mode, two fp32 LDS images chosen per operand by how the SOURCE is laid out (single buffer, register prefetch):
//   k-contiguous source ([rows,K] row-major, e.g. X or a Linear weight): LDS [row][36] (32 k + 4 pad), one ds_read_b128 per
//     32-row fragment per 8 k; pitch 36 makes every 16-lane b128 group hit 64 distinct banks.
//   m-contiguous source ([K,rows], e.g. dY in dW = dY^T X): LDS [k][rows], ds_read_b32 (lanes read 32 consecutive floats).
//   The MFMA k-slot of lane-half `hi` at step (t4,u) is k = 8*t4 + 4*hi + u for BOTH operands; the contraction order is a
//   permutation of 0..31, which fp32 accumulation does not care about.
// bf16x3 mode, two bf16 planes (hi, lo) per operand tile, double-buffered (one barrier per chunk):
//   k-contiguous source: unpadded [row][32] planes with XOR-swizzled 16-byte units, ds_read_b128 fragments;
//   m-contiguous source: [k][rows+32] planes in source orientation, fragments through the LDS transpose read ds_read_b64_tr_b16.
#include "gemm_core.h"

// the plane-fed LDS-DMA kernels live in their own translation units
int advmil_launch_nt_planes(int tile, bool a_single, dim3 pgrid, hipStream_t stream, const GemmArgs& g);
int advmil_launch_tn_planes(int tile, bool b_single, dim3 tgrid, hipStream_t stream, const GemmArgs& g);


// WR x WC waves per workgroup (2x2 = the 256-thread tiles; 4x2 = the 512-thread 256x192 / 256x128 tiles of the bf16x3 variant, whose
// time is set by how many operand bytes a CU pulls through its L1 per flop: ~11 B/clk/CU whatever the inner loop looks like).
// LDS floats of one workgroup of gemm_f32_body (operand buffers; the epilogue's per-wave areas reuse them)
template <bool A_KC, bool B_KC, int TM, int TN, bool SPLIT, int BKT, int WR, int WC>
__host__ __device__ constexpr int gemm_smem_floats() {
  constexpr int BM_ = 32 * TM * WR, BN_ = 32 * TN * WC, PITCH = BKT + 4;
  constexpr int TILEF_A = !SPLIT ? BM_ * PITCH : (A_KC ? BM_ * PITCH_PS(BKT) : BKT * PITCH_MC(BM_));
  constexpr int TILEF_B = !SPLIT ? BN_ * PITCH : (B_KC ? BN_ * PITCH_PS(BKT) : BKT * PITCH_MC(BN_));
  constexpr int NBUF = SPLIT ? 2 : 1;
  constexpr int PATCH_FLOATS = WR * WC * EPI_WAVE_FLOATS(TM, TN);
  return NBUF * (TILEF_A + TILEF_B) > PATCH_FLOATS ? NBUF * (TILEF_A + TILEF_B) : PATCH_FLOATS;
}

// The workgroup's work: tile `bid` (XCD-aware order), split `z`. A device function so that the grouped launch below (several small
// contractions in ONE launch) runs the same code as the plain kernel.
template <bool A_KC, bool B_KC, int TM, int TN, bool SPLIT, int PRE, int BKT, int WR, int WC>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, const int bid, const int z, float* const smem) {
  constexpr int NT = 64 * WR * WC;
  constexpr int BM_ = 32 * TM * WR, BN_ = 32 * TN * WC;
  // BKT = k per chunk, 32 everywhere. Measured alternatives: 64 for the bf16x3 variant (no gain, +60 VGPRs); 64/128 for 64x64
  // tiles on launch-bound shapes (no gain: their cost was an epilogue array in scratch memory, not the K walk).
  constexpr int PITCH = BKT + 4;
  // bf16x3: k-contiguous operands are kept pre-split in LDS (two bf16 planes, 160 B per row instead of 144 B of fp32)
  // (MC operands: two [k][rows+32] bf16 planes read with the LDS transpose read)
  constexpr int TILEF_A = !SPLIT ? BM_ * PITCH : (A_KC ? BM_ * PITCH_PS(BKT) : BKT * PITCH_MC(BM_));   // floats per operand tile
  constexpr int TILEF_B = !SPLIT ? BN_ * PITCH : (B_KC ? BN_ * PITCH_PS(BKT) : BKT * PITCH_MC(BN_));
  // bf16x3: two LDS buffers (one barrier per chunk, the next chunk is staged while the current one feeds the matrix pipe).
  // The exact variant stays single-buffered: there the doubled LDS footprint costs co-resident workgroups and measured slower.
  constexpr int NBUF = SPLIT ? 2 : 1;
  constexpr int BUF_FLOATS = TILEF_A + TILEF_B;
  static_assert(gemm_smem_floats<A_KC, B_KC, TM, TN, SPLIT, BKT, WR, WC>() >= NBUF * BUF_FLOATS, "LDS size helper out of step");
  float* const sA = smem;
  float* const sB = smem + TILEF_A;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  // XCD-aware tile order. Workgroup b runs on XCD b % 8 (each XCD has a private 4 MB L2), so tiles that share an
  // operand panel are given ids 8 apart: same XCD, dispatched back to back -> the panel is fetched from HBM once and
  // re-read from that L2. "inner" is the shorter tile axis: n-tiles of one A row-panel for the forward GEMMs
  // (ntiles = 2..4), m-tiles of one B panel for the dW = dY^T X contractions (mtiles = 3..6).
  int mt_i, nt_i;
  {
    const bool inner_n = g.ntiles <= g.mtiles;
    const int inner = inner_n ? g.ntiles : g.mtiles, outer = inner_n ? g.mtiles : g.ntiles;
    const int per_group = 8 * inner, full = (outer / 8) * per_group;
    int o, i_;
    if (bid < full) {
      const int r = bid % per_group;
      o = (bid / per_group) * 8 + (r & 7);
      i_ = r >> 3;
    } else {
      const int rem = outer - (outer / 8) * 8, r = bid - full;
      o = (outer / 8) * 8 + r % rem;
      i_ = r / rem;
    }
    mt_i = inner_n ? o : i_;
    nt_i = inner_n ? i_ : o;
  }
  const int64_t m0 = (int64_t)mt_i * BM_, n0 = (int64_t)nt_i * BN_;
  const int64_t kbeg = (int64_t)z * g.k_chunk;
  const int64_t kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  if constexpr (SPLIT) {
    // ---- bf16x3 main loop, LDS double-buffered; both operands live pre-split (hi/lo bf16 planes) in LDS
    OperandStage<A_KC, BM_, BKT, (PRE & 1) != 0, NT> ra;
    OperandStage<B_KC, BN_, BKT, (PRE & 2) != 0, NT> rb;
    const bf16raw* const a_hi = reinterpret_cast<const bf16raw*>(g.epi.a_hi);
    const bf16raw* const a_lo = reinterpret_cast<const bf16raw*>(g.epi.a_lo);
    const bf16raw* const b_hi = reinterpret_cast<const bf16raw*>(g.epi.b_hi);
    const bf16raw* const b_lo = reinterpret_cast<const bf16raw*>(g.epi.b_lo);
    auto fetch = [&](int64_t k0) {
      ra.load(g.A, a_hi, a_lo, g.lda, m0, g.M, k0, kend, tid);
      rb.load(g.B, b_hi, b_lo, g.ldb, n0, g.N, k0, kend, tid);
    };
    auto stage = [&](float* nA, float* nB) {
      ra.store(reinterpret_cast<bf16raw*>(nA), tid);
      rb.store(reinterpret_cast<bf16raw*>(nB), tid);
    };
    int cur = 0;
    if (kbeg < kend) {
      fetch(kbeg);
      stage(sA, sB);
      if (kbeg + BKT < kend) fetch(kbeg + BKT);
      __syncthreads();
    }
    for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
      const bf16raw* cA = reinterpret_cast<const bf16raw*>(sA + cur * BUF_FLOATS);
      const bf16raw* cB = reinterpret_cast<const bf16raw*>(sB + cur * BUF_FLOATS);
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          if (A_KC) read_frag_presplit<BM_, BKT>(cA, wr * 32 * TM + a * 32, ks, i, hi, ah[a], al[a]);
          else read_frag_presplit_mc<BM_, BKT>(cA, wr * 32 * TM + a * 32, ks, lane, ah[a], al[a]);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if (B_KC) read_frag_presplit<BN_, BKT>(cB, wc * 32 * TN + b * 32, ks, i, hi, bh[b], bl[b]);
          else read_frag_presplit_mc<BN_, BKT>(cB, wc * 32 * TN + b * 32, ks, lane, bh[b], bl[b]);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
          }
        // stage chunk c+1 behind the first MFMA group, then request chunk c+2. (Interleaving the staging slices between the
        // MFMA triples at source level was measured: no gain on 128x128, -15% on 64x64 tiles.)
        if (ks == 0 && k0 + BKT < kend) {
          stage(sA + (cur ^ 1) * BUF_FLOATS, sB + (cur ^ 1) * BUF_FLOATS);
          if (k0 + 2 * BKT < kend) fetch(k0 + 2 * BKT);
        }
      }
      __syncthreads();   // buffer cur^1 complete, buffer cur free
      cur ^= 1;
    }
  } else {
    // ---- exact fp32 main loop, single LDS buffer
    float4 ra[BM_ * BKT / (4 * NT)], rb[BN_ * BKT / (4 * NT)];
    if (kbeg < kend) {
      load_tile<A_KC, BM_, BKT, NT>(g.A, g.lda, m0, g.M, kbeg, kend, tid, ra);
      load_tile<B_KC, BN_, BKT, NT>(g.B, g.ldb, n0, g.N, kbeg, kend, tid, rb);
    }
    for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
      __syncthreads();  // all waves finished reading the previous chunk
      store_tile<A_KC, BM_, BKT, NT>(sA, tid, ra);
      store_tile<B_KC, BN_, BKT, NT>(sB, tid, rb);
      __syncthreads();
      if (k0 + BKT < kend) {  // prefetch next chunk; lands while the MFMAs below run
        load_tile<A_KC, BM_, BKT, NT>(g.A, g.lda, m0, g.M, k0 + BKT, kend, tid, ra);
        load_tile<B_KC, BN_, BKT, NT>(g.B, g.ldb, n0, g.N, k0 + BKT, kend, tid, rb);
      }
#pragma unroll
      for (int t4 = 0; t4 < BKT / 8; ++t4) {
        float fa[TM][4], fb[TN][4];
#pragma unroll
        for (int a = 0; a < TM; ++a) read_frag<A_KC, BM_, BKT>(sA, wr * 32 * TM + a * 32, t4, i, hi, fa[a]);
#pragma unroll
        for (int b = 0; b < TN; ++b) read_frag<B_KC, BN_, BKT>(sB, wc * 32 * TN + b * 32, t4, i, hi, fb[b]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][u], fb[b][u], acc[a][b], 0, 0, 0);
      }
    }
  }

  gemm_epilogue<TM, TN, WR, WC>(g, acc, smem, wave, lane, wr, wc, m0, n0, z, nt_i);
}

template <bool A_KC, bool B_KC, int TM, int TN, bool SPLIT, int PRE, int BKT, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC > 4 || TM * TN > 4 ? 1 : 2)) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<A_KC, B_KC, TM, TN, SPLIT, BKT, WR, WC>()];
  gemm_f32_body<A_KC, B_KC, TM, TN, SPLIT, PRE, BKT, WR, WC>(g, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// Up to GEMM_GROUP_MAX independent TN contractions of the 64x64 tile in ONE launch (gridDim.z = member): the weight gradients behind a
// fused small network are each a handful of tiles over a deep K -- 2-8 tiles x splits, 9-40 us apiece as separate launches because
// none fills the chip and each pays its own launch + K-walk latency; side by side they overlap.
#define GEMM_GROUP_MAX 4
struct GemmGroup {
  GemmArgs a[GEMM_GROUP_MAX];
};
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(GemmGroup gg) {
  __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<false, false, 1, 1, SPLIT, 32, 2, 2>()];
  const int m = (int)blockIdx.z, bid = (int)blockIdx.x, z = (int)blockIdx.y;
  // (a static member index per branch: the argument block stays in the kernel-argument segment, read through scalar loads)
#define GROUP_MEMBER(i)                                                                      \
  if (m == i) {                                                                              \
    if (bid < gg.a[i].mtiles * gg.a[i].ntiles && z < gg.a[i].splits)                         \
      gemm_f32_body<false, false, 1, 1, SPLIT, 0, 32, 2, 2>(gg.a[i], bid, z, smem);         \
    return;                                                                                  \
  }
  GROUP_MEMBER(0)
  GROUP_MEMBER(1)
  GROUP_MEMBER(2)
  GROUP_MEMBER(3)
#undef GROUP_MEMBER
}

// split-K reduction + epilogue; one thread per 4 consecutive columns
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g) {
  const int64_t n4 = g.N / 4;
  const int64_t total = g.M * n4;
  const advmil_epilogue_t& e = g.epi;
  uint64_t key = 0;
  float inv_keep = 1.0f;
  if (e.seed && e.drop_p > 0.0f) {
    key = rng_key(*e.seed, e.stream_id);
    inv_keep = hw_rcp(1.0f - e.drop_p);
  }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / n4, n = (idx % n4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = g.ws + m * g.N + n;
    const int64_t zs = g.M * g.N;
    int z = 0;
    for (; z + 8 <= g.splits; z += 8) {          // 8 partial loads in flight (the walk over 32-96 splits is latency-bound)
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = *reinterpret_cast<const float4*>(wp + (int64_t)(z + u) * zs);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += p[u].x; s.y += p[u].y; s.z += p[u].z; s.w += p[u].w; }
    }
    for (; z < g.splits; ++z) {
      const float4 p = *reinterpret_cast<const float4*>(wp + (int64_t)z * zs);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    float v[4] = {s.x, s.y, s.z, s.w};
    float* c = g.C + m * g.ldc + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o = epilogue_elem(e, v[q], m, n + q, g.N, key, inv_keep);
      if (e.accumulate) o += c[q];
      c[q] = o;
      v[q] = o;
    }
    if (e.c_hi) emit_planes4(e, m * g.ldc + n, v, 4);
  }
}

// The reduce of a split-K launch. A weight gradient accumulated into the optimizer's arena has the trivial epilogue (C += sum of the
// partial tiles, C dense): that sum goes through the merge queue (sumq.hip), i.e. while the stream is in deferral it shares ONE launch
// with the backward's other partial sums.
static int launch_splitk_reduce(const GemmArgs& g, hipStream_t stream) {
  const advmil_epilogue_t& e = g.epi;
  const bool plain = e.accumulate && e.alpha == 1.0f && !e.bias && !e.rowv && !e.maskref && e.act0 == 0 && e.act1 == 0 &&
                     !(e.seed && e.drop_p > 0.0f) && !e.c_hi && !e.gate_wc && g.ldc == g.N && (g.N & 3) == 0 &&
                     (((uintptr_t)g.C) & 15) == 0 && g.splits <= 64;
  if (plain) return advmil_sumq(stream, g.ws, g.splits, g.M * g.N, g.M * g.N, g.C, 1);
  const int64_t total = g.M * (g.N / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" size_t advmil_gemm_f32_workspace_bytes(int64_t M, int64_t N, int splits) {
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-bf16 ("bf16x3") on the bf16 matrix pipe
static int g_gemm_mode = 0;
static int g_nt_planes = []() { const char* e = getenv("ADVMIL_NT_PLANES"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int advmil_set_gemm_mode(int mode) {
  if (mode != 0 && mode != 1) return ADVMIL_EINVAL;
  g_gemm_mode = mode;
  return ADVMIL_OK;
}
extern "C" int advmil_get_gemm_mode(void) { return g_gemm_mode; }

template <int TM, int TN, bool SPLIT, int PRE, int WR, int WC>
static void launch_tile_m(int a_kc, int b_kc, dim3 grid, hipStream_t stream, const GemmArgs& g) {
  dim3 block(64 * WR * WC);
  if (a_kc && b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<true, true, TM, TN, SPLIT, PRE, 32, WR, WC>), grid, block, 0, stream, g);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false, TM, TN, SPLIT, PRE, 32, WR, WC>), grid, block, 0, stream, g);
  else if (!a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<false, false, TM, TN, SPLIT, PRE, 32, WR, WC>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<false, true, TM, TN, SPLIT, PRE, 32, WR, WC>), grid, block, 0, stream, g);
}

// PLANES: this tile is also built for operands that arrive as bf16 planes (the slab-sized contractions only use 22/12/11)
template <int TM, int TN, bool PLANES>
static void launch_tile(int a_kc, int b_kc, dim3 grid, hipStream_t stream, const GemmArgs& g, int pre) {
  if (g_gemm_mode != 1) { launch_tile_m<TM, TN, false, 0, 2, 2>(a_kc, b_kc, grid, stream, g); return; }
  if constexpr (PLANES) {
    switch (pre) {
      case 1: launch_tile_m<TM, TN, true, 1, 2, 2>(a_kc, b_kc, grid, stream, g); return;
      case 2: launch_tile_m<TM, TN, true, 2, 2, 2>(a_kc, b_kc, grid, stream, g); return;
      case 3: launch_tile_m<TM, TN, true, 3, 2, 2>(a_kc, b_kc, grid, stream, g); return;
      default: break;
    }
  }
  launch_tile_m<TM, TN, true, 0, 2, 2>(a_kc, b_kc, grid, stream, g);
}

// Which operands can be taken from caller-provided planes: both planes present, 16-byte aligned, pitch and contiguous extent
// multiples of 8 halfwords (a 16-byte piece never straddles a row end or the K range).
// lo == NULL with hi present: a SINGLE-plane operand -- the tensor IS bf16 (a bag stored in bf16: the x_storage = "bf16" mode), its
// lo plane is identically zero, nothing is fetched for it and the kernels built for it issue two MFMAs per product instead of three.
static int planes_usable(const void* hi, const void* lo, int64_t ld, int64_t contiguous_extent) {
  return hi && !((uintptr_t)hi & 15) && !((uintptr_t)lo & 15) && !(ld & 7) && !(contiguous_extent & 7);
}

// tile = 10*TM + TN  (22: 128x128, 23: 128x192, 13: 64x192, 12: 64x128, 11: 64x64).
static int64_t n_tiles(int tile, int64_t M, int64_t N) {
  const int tm = tile / 10, tn = tile % 10;
  return ((M + 64 * tm - 1) / (64 * tm)) * ((N + 64 * tn - 1) / (64 * tn));
}

// Launch plan, from the tools/gemm_sweep.py measurements on MI355X (256 CUs): a 4-wave workgroup alone on a CU leaves
// MFMA bubbles at every barrier, two or more co-resident workgroups fill them, so take the LARGEST tile that still
// yields >= 512 workgroups (8k-row bags -> 64x64 / 64x128 tiles at ~95-103 TF; 32k-row bags -> 128x192 at ~122 TF).
// If even 64x64 tiles are too few and K is deep (the dW = dY^T X contractions, K = bag length), split K so that
// ~768 workgroups each keep >= 1024 of K (partials reduced by a second launch).
static int plan_exact(int64_t M, int64_t N, int64_t K, int* tile, int* splits);

// Layout-aware plan. bf16x3 mode adds, from tools/gemm_slab_check.py on the 16 x 8k slab shapes (the swizzled LDS image made the
// 192-wide tiles fit two workgroups per CU; the 512-thread 256x192 tile stages 42% fewer operand bytes per flop):
//   N a multiple of 192, slab-sized M:  NT (both k-contiguous) -> 128x192 (gates 419 -> 385 us, embed FC 444 -> 421 us);
//                                      NN / TN               -> 256x192, 8 waves (dX 341 -> 314 us);
//   deep-K weight gradients whose [M,N] divides into 256x192 / 192x256 / 128x256 tiles -> that 8-wave tile with one workgroup per CU
//     (dWab 768x384: 356 -> 245 us).
extern "C" int advmil_gemm_f32_plan_layout(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, int* tile, int* splits) {
  if (!tile || !splits || M <= 0 || N <= 0 || K <= 0) return ADVMIL_EINVAL;
  if (g_gemm_mode == 1 && N % 192 == 0) {
    if (M >= 16384) {
      // NN / TN forms: the 8-wave 256x192 tile needs a full wave of workgroups; the 16384-row slab of a 2-bag step (128 of them) runs
      // faster on 128x128 tiles (dh 16384 x 384 x 768: 53 -> 48 us, tools/probe/bag2_shapes.py)
      const bool few = (M / 256) * (N / 192) < 256 && N % 128 == 0 && M % 128 == 0;
      *tile = (a_kc && b_kc) ? 23 : (few ? 22 : 43);
      *splits = 1;
      return ADVMIL_OK;
    }
  }
  if (g_gemm_mode == 1 && K >= 16384 && M < 16384 && !(a_kc && b_kc)) {      // deep-K weight gradients: one wave of 8-wave workgroups
    const int t8 = (M % 256 == 0 && N % 192 == 0) ? 43 : (M % 192 == 0 && N % 256 == 0) ? 34 : (M % 128 == 0 && N % 256 == 0) ? 24 : 0;
    if (t8) {
      const int64_t w = n_tiles(t8, M, N);
      int64_t sp = 256 / w;                      // one 512-thread workgroup per CU
      if (sp > K / 1024) sp = K / 1024;
      if (sp >= 2) { *tile = t8; *splits = (int)sp; return ADVMIL_OK; }
    }
  }
  return plan_exact(M, N, K, tile, splits);
}

// Tile of the plane-fed NT kernel for this shape (82 / 83), or 0 when the shape does not qualify (then the generic kernel runs).
extern "C" int advmil_gemm_f32_plan_planes(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, int* tile) {
  if (!tile) return ADVMIL_EINVAL;
  *tile = 0;
  if (g_gemm_mode != 1 || !g_nt_planes || !a_kc || !b_kc || M < 4096 || (M % 256) || (K % 32) || (N % 128)) return ADVMIL_OK;
  const char* force = getenv("ADVMIL_NT_PLANES_TN");
  // widest tile that divides N: most flops per staged byte. (A 256x256 form measured equal to 256x192 on every slab shape and, as a
  // persistent kernel, no longer fits the register file beside the streaming epilogue: not built.)
  int tnp = (N % 192 == 0) ? 3 : 2;
  // ... unless that leaves CUs without a tile while the 128-wide one does not (the 16384-row slab of a 2-bag step, N = 384: 128 tiles of
  // 256x192 against 192 of 256x128: 60.7 -> 48.4 us, tools/probe/bag2_shapes.py)
  if (tnp == 3 && N % 128 == 0 && (M / 256) * (N / 192) < 256 && (M / 256) * (N / 128) >= 192) tnp = 2;
  if (force && (force[0] == '2' || force[0] == '3') && N % (64 * (force[0] - '0')) == 0) tnp = force[0] - '0';
  if (K < 64 || (uint64_t)M * (uint64_t)K * 2 >= (1ull << 32) || (uint64_t)N * (uint64_t)K * 2 >= (1ull << 32)) return ADVMIL_OK;
  static const int min_tiles = []() { const char* e = getenv("ADVMIL_NT_PLANES_MIN_TILES"); return e ? atoi(e) : 256; }();
  // one 8-wave workgroup per CU: less than one full wave of tiles loses to the small tiles. (384 -- 1.5 waves -- until round 4; 256
  // measured +0.9 % on the PatchGCN step, whose 65536 x 128 layers are exactly one wave, and neutral at 1-5 ABMIL bags and ESAT 8k)
  const int64_t nt_ = (M / 256) * (N / (64 * tnp));
  if (nt_ < min_tiles && !(tnp == 2 && N % 192 == 0 && nt_ >= 192)) return ADVMIL_OK;
  *tile = 80 + tnp;
  return ADVMIL_OK;
}

// TN contraction of two operands held as planes (A[K,M], B[K,N]): tile code 91 / 92 / 93 (128x256 / 256x128 / 256x256) and split
// count of gemm_tn_planes_kernel, or tile 0 when the shape does not qualify. Splits: as many as keep every XCD's share of the
// workgroups (groups of tiles that share a row panel, see the kernel) within its 32 CUs, with >= 1024 of K per split.
extern "C" int advmil_gemm_f32_plan_tn_planes(int64_t M, int64_t N, int64_t K, int* tile, int* splits) {
  if (!tile || !splits) return ADVMIL_EINVAL;
  *tile = 0; *splits = 1;
  static const bool off = []() { const char* e = getenv("ADVMIL_TN_PLANES"); return e && e[0] == '0'; }();
  if (off || g_gemm_mode != 1 || (K % 32) || K < 8192 || (M % 128) || (N % 128)) return ADVMIL_OK;
  int t = 0;
  if (M % 256 == 0 && N % 256 == 0) t = 93;
  else if (N % 256 == 0) t = 91;
  else if (M % 256 == 0) t = 92;
  else return ADVMIL_OK;
  const int64_t mt = M / (t == 91 ? 128 : 256), nt = N / (t == 92 ? 128 : 256);
  const int64_t gs = M >= N ? nt : mt, og = M >= N ? mt : nt;
  if (gs > 32) return ADVMIL_OK;
  int64_t sp = (8 * (32 / gs)) / og;                    // groups per XCD x 8 XCDs, over the groups of one split
  // >= 512 of K per split (1024 until round 5: at the 16384 rows of a 2-bag step that left 64-192 workgroups for 256 CUs; dW_D 128 x 1024:
  // 57 -> 35 us, dW1 384 x 1024: 58 -> 53 us, dWab 768 x 384: 53 -> 46 us, tools/probe/bag2_shapes.py)
  static const int64_t mink = []() { const char* e = getenv("ADVMIL_TN_PLANES_MINK"); return (int64_t)(e ? atoi(e) : 512); }();
  if (sp > K / mink) sp = K / mink;
  if (sp < 1) return ADVMIL_OK;
  if (sp * mt * nt < 128) return ADVMIL_OK;             // fewer than half a wave of workgroups: the generic plan spreads better
  *tile = t; *splits = (int)sp;
  return ADVMIL_OK;
}

extern "C" int advmil_gemm_f32_plan(int64_t M, int64_t N, int64_t K, int* tile, int* splits) {
  return advmil_gemm_f32_plan_layout(1, 1, M, N, K, tile, splits);
}

static int plan_exact(int64_t M, int64_t N, int64_t K, int* tile, int* splits) {
  static const int order[5] = {22, 12, 23, 13, 11};   // 128x192 is never better than 128x128 / 64x128 once M is a slab (tools/gemm_slab_check.py)
  for (int c = 0; c < 5; ++c)
    if (n_tiles(order[c], M, N) >= 512) { *tile = order[c]; *splits = 1; return ADVMIL_OK; }
  const int64_t w11 = n_tiles(11, M, N);
  if (K >= 512 && w11 < 384 && (N & 3) == 0) {
    static const int sorder[3] = {22, 12, 11};
    if (g_gemm_mode == 1) {
      // bf16x3, deep K over a small [M, N] (region-level weight gradients, 384 x 384 x 32768 ...): the LARGEST of the 128x128 / 64x128
      // tiles that still gives ~1.5 waves of workgroups (>= 256 of them, <= 64 splits, >= 256 of K each) -- the 64x64 tile the rule
      // below would pick streams its operands at a third of the rate (tools/probe/splitk_sweep.py: 384x384x32768 97 -> 62 us,
      // 256x128x65536 46 -> 40 us, 1152x384x32768 145 -> 125 us)
      for (int c = 0; c < 2; ++c) {
        const int64_t w = n_tiles(sorder[c], M, N);
        int64_t sp = (384 + w - 1) / w;
        if (sp > 64) sp = 64;
        if (sp >= 2 && w * sp >= 256 && K / sp >= 256) { *tile = sorder[c]; *splits = (int)sp; return ADVMIL_OK; }
      }
    }
    for (int c = 0; c < 3; ++c) {
      const int64_t w = n_tiles(sorder[c], M, N);
      const int64_t sp = (768 + w - 1) / w;
      if (K / sp >= 1024) { *tile = sorder[c]; *splits = (int)sp; return ADVMIL_OK; }
    }
    // few 64x64 tiles over a long K (weight gradients of the [B,d] / region-level layers): a workgroup walking K/16 serially was
    // 15 us of pure latency; >= 256 k per workgroup and up to 64 partials (the reduce launch keeps 8 loads in flight)
    int64_t sp = (768 + w11 - 1) / w11;
    const int64_t cap = K / 256 > 0 ? K / 256 : 1;
    if (sp > cap) sp = cap;
    if (sp > 64) sp = 64;
    *tile = 11; *splits = (int)sp;
    return ADVMIL_OK;
  }
  *tile = 11; *splits = 1;
  return ADVMIL_OK;
}

// waves along N of a tile code: 2 for the 256-thread tiles and 43/42, 4 for the 2 x 4 wave grids 34/24
static int tile_wc(int tile) { return (tile == 34 || tile == 24) ? 4 : 2; }
extern "C" int advmil_gemm_f32_gate_blocks(int tile, int64_t N) {
  if (tile >= 82 && tile <= 84) return (int)(N / (64 * (tile % 10))) * 2;      // plane-fed NT kernel: 2 waves along N
  if (g_gemm_mode != 1) {
    if (tile / 10 == 4) tile = 20 + tile % 10;
    else if (tile % 10 == 4) tile = (tile / 10 == 3) ? 23 : 22;
  }
  const int tn = tile % 10;
  return (int)((N + 64 * tn - 1) / (64 * tn)) * tile_wc(tile);
}

// accumulator blocks per wave and wave grid of a generic tile code: (TM, TN, WR, WC); false for codes without a kernel
static bool tile_geom(int tile, int& tm, int& tn, int& wr, int& wc) {
  switch (tile) {
    case 43: tm = 2; tn = 3; wr = 4; wc = 2; return true;
    case 42: tm = 2; tn = 2; wr = 4; wc = 2; return true;
    case 34: tm = 3; tn = 2; wr = 2; wc = 4; return true;
    case 24: tm = 2; tn = 2; wr = 2; wc = 4; return true;
    case 23: case 22: case 13: case 12: case 11: tm = tile / 10; tn = tile % 10; wr = 2; wc = 2; return true;
    default: return false;
  }
}

extern "C" int64_t advmil_gemm_f32_colsum_rows(int tile, int64_t M, int64_t N) {
  int tm, tn, wr, wc;
  if (g_gemm_mode != 1 || !tile_geom(tile, tm, tn, wr, wc) || tm * tn < 4) return 0;
  const int64_t bm = 32 * tm * wr, bn = 32 * tn * wc;
  return ((M % bm) || (N % bn)) ? 0 : (M / bm) * wr;
}

extern "C" int advmil_merge_partials(const float* partial, int nblk, int64_t stride, int64_t ncols, float* out, int accumulate,
                                     advmil_stream_t stream) {
  if (!partial || !out || nblk <= 0 || ncols <= 0 || stride < ncols) return ADVMIL_EINVAL;
  return advmil_sumq((hipStream_t)stream, partial, nblk, stride, ncols, out, accumulate);
}

extern "C" int advmil_gemm_f32_tiled(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                     const float* B, int64_t ldb, float* C, int64_t ldc, const advmil_epilogue_t* epi,
                                     int splits, int tile, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !B || !epi || (!C && !epi->gate_wc && !epi->c_hi) || M <= 0 || N <= 0 || K <= 0) return ADVMIL_EINVAL;
  // C == NULL with c_hi / c_lo set: the operand planes of the result ONLY (a result that is consumed as a bf16x3 operand and nowhere else:
  // the ESAT in-projection feeding the attention kernels); one pass, nothing to accumulate into
  if (!C && !epi->gate_wc && (epi->accumulate || splits != 1 || epi->c2)) return ADVMIL_EINVAL;
  if ((lda & 3) || (ldb & 3)) return ADVMIL_EINVAL;
  if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N)) return ADVMIL_EINVAL;          // a row pitch shorter than the row it strides
  if (C && ldc < (epi->c2 ? (int64_t)epi->n_split : N)) return ADVMIL_EINVAL;      // (two-layer form: C holds the first n_split columns)
  if (a_kc ? (K & 3) : (M & 3)) return ADVMIL_EINVAL;
  if (b_kc ? (K & 3) : (N & 3)) return ADVMIL_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ADVMIL_EINVAL;
  if (splits < 1) splits = 1;
  const int64_t kchunks = (K + BK - 1) / BK;
  if (splits > kchunks) splits = (int)kchunks;
  if (splits > 1 && (N & 3)) return ADVMIL_EINVAL;
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.k_chunk = ((kchunks + splits - 1) / splits) * BK;
  splits = (int)((K + g.k_chunk - 1) / g.k_chunk);
  g.splits = splits;
  g.ws = (float*)ws;
  g.epi = *epi;
  if (splits > 1) {
    if (!ws || ws_bytes < advmil_gemm_f32_workspace_bytes(M, N, splits)) return ADVMIL_EWORKSPACE;
    if ((uintptr_t)ws & 15) return ADVMIL_EINVAL;
  }
  if (tile == 0) { int t = 0, sp = 0; advmil_gemm_f32_plan_layout(a_kc, b_kc, M, N, K, &t, &sp); tile = t; }
  if (g_gemm_mode != 1) {                                            // the 512-thread tiles exist for bf16x3 only
    if (tile / 10 == 4) tile = 20 + tile % 10;
    else if (tile % 10 == 4) tile = (tile / 10 == 3) ? 23 : 22;
  }
  const int tm = tile / 10, tn = tile % 10;
  g.mtiles = (int)((M + 64 * tm - 1) / (64 * tm));
  const int ntiles = (int)((N + 64 * tn - 1) / (64 * tn));
  g.ntiles = ntiles;
  dim3 grid(g.mtiles * ntiles, splits);
  int pre = 0;
  if (g_gemm_mode == 1) {
    // split-K chunks are multiples of BK = 32, so only the last chunk can end off a multiple of 8 -- covered by K % 8 == 0
    if (planes_usable(epi->a_hi, epi->a_lo, lda, a_kc ? K : M)) pre |= 1;
    if (planes_usable(epi->b_hi, epi->b_lo, ldb, b_kc ? K : N)) pre |= 2;
  }
  if ((epi->c_hi != nullptr) != (epi->c_lo != nullptr)) return ADVMIL_EINVAL;
  if (epi->maskbits) {    // the bit mask is read by the streaming epilogue only
    int tm_, tn_, wr_, wc_;
    if (g_gemm_mode != 1 || splits != 1 || !tile_geom(tile, tm_, tn_, wr_, wc_) || tm_ * tn_ < 4 || epi->gate_wc || epi->maskref || epi->accumulate ||
        (M % (32 * tm_ * wr_)) || (N % (32 * tn_ * wc_)) || (epi->act_split & 31) || epi->ldbits < N / 32 || (epi->seed && epi->drop_p > 0.0f && epi->rowv && epi->rng_row))
      return ADVMIL_EINVAL;
    // (everything else the streaming form asks for: a launch that fell back to the generic epilogue would silently ignore the bits)
    if ((C && ((((uintptr_t)C) & 15) || (ldc & 3))) || (((uintptr_t)epi->bias) & 15) || (epi->rowv && (((uintptr_t)epi->colv) & 15)) ||
        (epi->c_hi && ((((uintptr_t)epi->c_hi) | ((uintptr_t)epi->c_lo)) & 7)) || epi->c2)
      return ADVMIL_EINVAL;
  }
  if (epi->colsum) {      // per-wave column sums come out of the streaming epilogue only: whole tiles of a slab-sized tile, one pass
    int tm_, tn_, wr_, wc_;
    if (g_gemm_mode != 1 || splits != 1 || !tile_geom(tile, tm_, tn_, wr_, wc_) || tm_ * tn_ < 4 || epi->gate_wc || epi->accumulate ||
        (epi->seed && epi->drop_p > 0.0f))
      return ADVMIL_EINVAL;
    if ((M % (32 * tm_ * wr_)) || (N % (32 * tn_ * wc_)) || (((uintptr_t)epi->colsum) & 15) || (epi->act_split & 31)) return ADVMIL_EINVAL;
  }
  if (epi->gate_wc) {       // fused gate score: no split-K, no dropout, whole float4 column groups, one partial per 32*TN*... block
    if (splits != 1 || !epi->gate_out || (N & 3) || epi->drop_p > 0.0f) return ADVMIL_EINVAL;
    if (epi->gate_np != advmil_gemm_f32_gate_blocks(tile, N)) return ADVMIL_EINVAL;
  }
  // NT form with both operands as planes: the LDS-DMA kernel (tile codes 82 / 83 = 256 x 128 / 192, 8 waves). The plan
  // (advmil_gemm_f32_plan_planes, or tile 0 here) picks it whenever the shape qualifies; ADVMIL_NT_PLANES=0 turns it off.
  if (tile == 0 && pre == 3 && splits == 1) { int t = 0; advmil_gemm_f32_plan_planes(a_kc, b_kc, M, N, K, &t); if (t) tile = t; }
  if (tile >= 82 && tile <= 86) {
    // 86: 256x128 with the PLAIN streaming epilogue (like 85): the two-layer launch of a slab too short to fill the chip with 256x256 tiles
    const int tnp = tile == 85 ? 4 : tile == 86 ? 2 : tile % 10, bm = 256, bkt = 32;
    const bool plain = tile == 85 || tile == 86;
    if (tile == 84 && !epi->gate_wc) return ADVMIL_EINVAL;        // 256x256: the fused gate score only
    if (plain && (epi->gate_wc || epi->rowv || epi->maskref || epi->accumulate || (epi->seed && epi->drop_p > 0.0f))) return ADVMIL_EINVAL;
    if (epi->c2) {      // two layers in one launch: the plain forms only, split on a 32-column boundary inside N
      if (!plain || epi->n_split <= 0 || epi->n_split >= N || (epi->n_split & 31) || (epi->ldc2 & 3) || ((uintptr_t)epi->c2 & 15) ||
          epi->act_split != epi->n_split)
        return ADVMIL_EINVAL;
    }
    if (g_gemm_mode != 1 || !a_kc || !b_kc || pre != 3 || splits != 1 || (M % bm) || (K % bkt) || (N % (64 * tnp))) return ADVMIL_EINVAL;
    if (epi->gate_wc && (!epi->gate_out || epi->drop_p > 0.0f || epi->gate_np != advmil_gemm_f32_gate_blocks(tile, N))) return ADVMIL_EINVAL;
    if (K < 64) return ADVMIL_EINVAL;                   // the three-slot ring prefetches two chunks ahead, across tiles
    if ((uint64_t)M * (uint64_t)lda * 2 >= (1ull << 32) || (uint64_t)N * (uint64_t)ldb * 2 >= (1ull << 32)) return ADVMIL_EINVAL;   // 32-bit plane offsets
    g.mtiles = (int)(M / bm);
    g.ntiles = (int)(N / (64 * tnp));
    static const int ncu = []() { int dev = 0, n = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const int ntile_all = g.mtiles * g.ntiles;
    dim3 pgrid(ntile_all < ncu ? ntile_all : ncu);        // persistent: one workgroup per CU walks its share of the tiles
    if (!epi->b_lo) return ADVMIL_EINVAL;                 // (a single-plane B has no instantiation here: weights are always hi + lo)
    return advmil_launch_nt_planes(tile, !epi->a_lo, pgrid, stream, g);
  }
  if (tile >= 91 && tile <= 93) {
    // TN over caller-held planes of both operands (tile codes 91 / 92 / 93 = 128x256 / 256x128 / 256x256): the LDS-DMA kernel
    const int bm = tile == 91 ? 128 : 256, bn = tile == 92 ? 128 : 256;
    if (g_gemm_mode != 1 || a_kc || b_kc || pre != 3 || (M % bm) || (N % bn) || (K % 32) || (g.k_chunk % 32)) return ADVMIL_EINVAL;
    if (g.k_chunk / 32 < 3 || epi->gate_wc || epi->c2) return ADVMIL_EINVAL;
    if ((uint64_t)32 * (uint64_t)lda * 2 + (uint64_t)M * 2 >= (1ull << 32) || (uint64_t)32 * (uint64_t)ldb * 2 + (uint64_t)N * 2 >= (1ull << 32)) return ADVMIL_EINVAL;
    g.mtiles = (int)(M / bm);
    g.ntiles = (int)(N / bn);
    const int gs = M >= N ? g.ntiles : g.mtiles, og = M >= N ? g.mtiles : g.ntiles;
    const dim3 tgrid((unsigned)(8 * ((splits * og + 7) / 8) * gs));
    if (!epi->a_lo) return ADVMIL_EINVAL;                 // (only B -- the slab -- may be a single-plane operand here)
    { const int rc = advmil_launch_tn_planes(tile, !epi->b_lo, tgrid, stream, g); if (rc) return rc; }
    if (splits > 1) {
      const int rc = launch_splitk_reduce(g, stream);
      if (rc) return rc;
    }
    return ADVMIL_OK;
  }
  switch (tile) {
    case 23: launch_tile<2, 3, false>(a_kc, b_kc, grid, stream, g, pre); break;
    case 22: launch_tile<2, 2, true>(a_kc, b_kc, grid, stream, g, pre); break;
    case 13: launch_tile<1, 3, false>(a_kc, b_kc, grid, stream, g, pre); break;
    case 12: launch_tile<1, 2, true>(a_kc, b_kc, grid, stream, g, pre); break;
    case 11: launch_tile<1, 1, true>(a_kc, b_kc, grid, stream, g, pre); break;
    case 43:                                                                               // 256x192, 8 waves
      // A operand from caller-held planes (dG written as planes by the gate backward): dh = dG Wab (NN), dWab = dG^T h (TN)
      if (pre == 1 && a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, false, 2, 3, true, 1, 32, 4, 2>), grid, dim3(512), 0, stream, g);
      else if (pre == 1 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2, 3, true, 1, 32, 4, 2>), grid, dim3(512), 0, stream, g);
      else launch_tile_m<2, 3, true, 0, 4, 2>(a_kc, b_kc, grid, stream, g);
      break;
    case 42: launch_tile_m<2, 2, true, 0, 4, 2>(a_kc, b_kc, grid, stream, g); break;   // 256x128, 8 waves
    // 192x256 / 128x256, 8 waves (2 x 4): the weight-gradient contractions dY^T X over the slab rows. With the slab's planes at
    // hand (B = X: 3/4 to 9/10 of the staged elements) that operand is staged without any conversion work.
    case 34:
      if (pre == 2 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 3, 2, true, 2, 32, 2, 4>), grid, dim3(512), 0, stream, g);
      else if (pre == 3 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 3, 2, true, 3, 32, 2, 4>), grid, dim3(512), 0, stream, g);   // dY as planes too
      else launch_tile_m<3, 2, true, 0, 2, 4>(a_kc, b_kc, grid, stream, g);
      break;
    case 24:
      if (pre == 2 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2, 2, true, 2, 32, 2, 4>), grid, dim3(512), 0, stream, g);
      else if (pre == 3 && !a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2, 2, true, 3, 32, 2, 4>), grid, dim3(512), 0, stream, g);
      else launch_tile_m<2, 2, true, 0, 2, 4>(a_kc, b_kc, grid, stream, g);
      break;
    default: return ADVMIL_EINVAL;
  }
  ADVMIL_LAUNCH_CHECK();
  if (splits > 1) {
    const int rc = launch_splitk_reduce(g, stream);
    if (rc) return rc;
  }
  return ADVMIL_OK;
}

extern "C" int advmil_gemm_f32(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                               const float* B, int64_t ldb, float* C, int64_t ldc, const advmil_epilogue_t* epi,
                               int splits, void* ws, size_t ws_bytes, advmil_stream_t stream) {
  return advmil_gemm_f32_tiled(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, epi, splits, 0, ws, ws_bytes, stream);
}

// ---- several small deep-K TN contractions C_i (+)= A_i^T B_i as ONE launch (gemm_tn_group_kernel): 64x64 tiles, split-K per member,
// partial tiles merged by the common reduce (deferrable: advmil_defer_sums). Same kernel body and k order as the plain launch of the
// 64x64 tile with the same split count -> bit-identical to it.
static int group_member_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t w = ((M + 63) / 64) * ((N + 63) / 64);
  int64_t sp = K / 256;                      // >= 256 of K per workgroup (tools/probe/chain_wgrad_sweep.py: flat from K/512 to K/128)
  if (sp > 64) sp = 64;
  if (sp * w > 1024) sp = 1024 / w;
  return sp < 1 ? 1 : (int)sp;
}
static size_t group_member_ws(const advmil_gemm_tn_call_t& c) {
  const int sp = group_member_splits(c.M, c.N, c.K);
  const int64_t kchunks = (c.K + BK - 1) / BK;
  const int64_t kc = ((kchunks + sp - 1) / sp) * BK;
  const int splits = (int)((c.K + kc - 1) / kc);
  return (advmil_gemm_f32_workspace_bytes(c.M, c.N, splits) + 255) & ~(size_t)255;
}
extern "C" size_t advmil_gemm_tn_group_workspace_bytes(const advmil_gemm_tn_call_t* calls, int n) {
  if (!calls || n < 1 || n > GEMM_GROUP_MAX) return 0;
  size_t t = 0;
  for (int i = 0; i < n; ++i) t += group_member_ws(calls[i]);
  return t;
}
extern "C" int advmil_gemm_tn_group(const advmil_gemm_tn_call_t* calls, int n, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!calls || n < 1 || n > GEMM_GROUP_MAX) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_gemm_tn_group_workspace_bytes(calls, n)) return ADVMIL_EWORKSPACE;
  if (ws_bytes && (!ws || ((uintptr_t)ws & 15))) return ADVMIL_EINVAL;
  GemmGroup gg{};                              // (zero epilogue blocks: no bias / activation / dropout / planes)
  unsigned gx = 1, gy = 1;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    const advmil_gemm_tn_call_t& c = calls[i];
    if (!c.A || !c.B || !c.C || c.M <= 0 || c.N <= 0 || c.K <= 0 || (c.M & 3) || (c.N & 3) || (c.lda & 3) || (c.ldb & 3) || c.lda < c.M || c.ldb < c.N ||
        c.ldc < c.N || (((uintptr_t)c.A | (uintptr_t)c.B) & 15))
      return ADVMIL_EINVAL;
    GemmArgs& g = gg.a[i];
    g.M = c.M; g.N = c.N; g.K = c.K; g.A = c.A; g.lda = c.lda; g.B = c.B; g.ldb = c.ldb; g.C = c.C; g.ldc = c.ldc;
    const int sp = group_member_splits(c.M, c.N, c.K);
    const int64_t kchunks = (c.K + BK - 1) / BK;
    g.k_chunk = ((kchunks + sp - 1) / sp) * BK;
    g.splits = (int)((c.K + g.k_chunk - 1) / g.k_chunk);
    g.ws = g.splits > 1 ? (float*)((char*)ws + off) : nullptr;
    off += group_member_ws(c);
    g.mtiles = (int)((c.M + 63) / 64);
    g.ntiles = (int)((c.N + 63) / 64);
    g.epi.alpha = 1.0f;
    g.epi.act_split = 1 << 30;
    g.epi.accumulate = c.accumulate ? 1 : 0;
    if ((unsigned)(g.mtiles * g.ntiles) > gx) gx = (unsigned)(g.mtiles * g.ntiles);
    if ((unsigned)g.splits > gy) gy = (unsigned)g.splits;
  }
  if (g_gemm_mode == 1) hipLaunchKernelGGL((gemm_tn_group_kernel<true>), dim3(gx, gy, n), dim3(256), 0, stream, gg);
  else hipLaunchKernelGGL((gemm_tn_group_kernel<false>), dim3(gx, gy, n), dim3(256), 0, stream, gg);
  ADVMIL_LAUNCH_CHECK();
  for (int i = 0; i < n; ++i)
    if (gg.a[i].splits > 1) {
      const int rc = launch_splitk_reduce(gg.a[i], stream);
      if (rc) return rc;
    }
  return ADVMIL_OK;
}

// ---- fp32 matrix -> bf16 planes (hi = bf16(x), lo = bf16(x - hi)); the same rounding the staging path applies on the fly
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t n, bf16raw* __restrict__ hi,
                                                           bf16raw* __restrict__ lo) {
  const int64_t n8 = n >> 3;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n8; idx += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * idx], b = reinterpret_cast<const float4*>(src)[2 * idx + 1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    union { __bf16 v[8]; uint4 u; } h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h.v[j] = (__bf16)x[j];
      l.v[j] = (__bf16)(x[j] - (float)h.v[j]);
    }
    reinterpret_cast<uint4*>(hi)[idx] = h.u;
    if (lo) reinterpret_cast<uint4*>(lo)[idx] = l.u;         // lo == NULL: the bf16 rounding alone (a bag entering a bf16 slab)
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t j = (n8 << 3) + threadIdx.x;
    const __bf16 h = (__bf16)src[j];
    const __bf16 l = (__bf16)(src[j] - (float)h);
    hi[j] = *reinterpret_cast<const bf16raw*>(&h);
    if (lo) lo[j] = *reinterpret_cast<const bf16raw*>(&l);
  }
}

extern "C" int advmil_split_planes(const float* src, int64_t n, void* hi, void* lo, advmil_stream_t stream_) {
  if (!src || !hi || n < 0) return ADVMIL_EINVAL;            // (lo may be NULL: hi = bf16(x) only)
  if (((uintptr_t)src & 15) || ((uintptr_t)hi & 15) || ((uintptr_t)lo & 15)) return ADVMIL_EINVAL;
  if (n == 0) return ADVMIL_OK;
  int64_t blocks = ((n >> 3) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, src, n, (bf16raw*)hi, (bf16raw*)lo);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

