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
int advmil_launch_splitk_reduce(const GemmArgs& g, hipStream_t stream) {
  const advmil_epilogue_t& e = g.epi;
  const bool plain = e.accumulate && e.alpha == 1.0f && !e.bias && !e.rowv && !e.maskref && e.act0 == 0 && e.act1 == 0 &&
                     !(e.seed && e.drop_p > 0.0f) && !e.c_hi && !e.gate_wc && g.ldc == g.N && (g.N & 3) == 0 &&
                     (((uintptr_t)g.C) & 15) == 0 && g.splits <= 64;
  if (plain) return advmil_sumq(stream, g.ws, g.splits, g.M * g.N, g.M * g.N, g.C, 1, nullptr, 0, nullptr, 0, e.c_rows_pair32 ? g.N : 0);
  if (e.c_rows_pair32) return ADVMIL_EINVAL;      // (rows in pair-block order: only the merge un-permutes them)
  const int64_t total = g.M * (g.N / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern int g_gemm_mode;            // gemm_host.hip

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

// the generic kernels' side of the dispatch (advmil_gemm_f32_tiled, gemm_host.hip, has checked the arguments and made the plan)
int advmil_launch_generic(int tile, int a_kc, int b_kc, int pre, dim3 grid, hipStream_t stream, const GemmArgs& g) {
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
  return ADVMIL_OK;
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
      const int rc = advmil_launch_splitk_reduce(gg.a[i], stream);
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

