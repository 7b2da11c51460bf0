#include "gemm_core.h"

// =====================================================================================
// TN contraction over operands that live in HBM as bf16 planes, both m-contiguous: C[M,N] (+)= A^T B with A[K,M], B[K,N] -- the
// deep-K weight gradients dW = dY^T X over the slab's rows (K = the slab's rows, M x N = the weight's shape; dY and X both already
// exist as planes: ops.act_dropout_bwd(planes_only), the gate backward's dG, the resident slab). Same idea as the NT plane kernel:
// nothing to convert, so a k-chunk goes global -> LDS by LDS-DMA and the inner loop is LDS transpose reads + MFMA only.
//   LDS image of an operand tile = the SOURCE orientation, dense: [32 k rows][BR m] bf16 (BR = 128 / 256 -> 256 / 512 bytes per
//   row), hi plane then lo plane; read with ds_read_b64_tr_b16 (lane map of read_frag_presplit_mc). A transpose read touches 4
//   consecutive k rows x one 64-byte window: with a dense pitch those four windows would fall into the same bank quarter, so the
//   64-byte windows of row k are stored XOR (k & 3) -- applied to the DMA's per-lane SOURCE address (the DMA writes lane-linear).
//   Workgroup = WR x WC waves of 64 x (32 TN) accumulator blocks; split-K over gridDim: partial tiles go to the workspace and the
//   common reduce launch applies the epilogue. Workgroup id -> (split, tile) so that the tiles of ONE split -- which share its A and
//   B row panels -- run on one XCD (id % 8) back to back.
//   Same k order inside a chunk and across chunks as gemm_f32_kernel's bf16x3 loop -> bit-identical partials.
// =====================================================================================
template <int BR>
__device__ __forceinline__ void read_frag_tn(const unsigned char* __restrict__ tile, int rbase, int ks, int lane, bf16x8& h, bf16x8& l) {
  constexpr int ROWB = BR * 2;
  const int q4 = (lane & 15) >> 2, b = (lane >> 4) & 1, e = lane & 3;
  const int krow = ks * 16 + (lane >> 5) * 8 + q4;                         // (krow & 3) == q4; the second read's row krow + 4 too
  const unsigned char* p = tile + krow * ROWB + (((rbase >> 5) ^ q4) << 6) + 32 * b + 8 * e;
  union { bf16x4_t q[2]; bf16x8 v; } x, y;
  x.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p));
  x.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + 4 * ROWB));
  y.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + 32 * ROWB));
  y.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + 36 * ROWB));
  h = x.v; l = y.v;
}

// BLO = 0: B is a single-plane (bf16) operand (the slab X of the x_storage = "bf16" mode): no B lo pieces, two MFMAs per product.
template <int BR>
__device__ __forceinline__ void read_frag_tn_hi(const unsigned char* __restrict__ tile, int rbase, int ks, int lane, bf16x8& h) {
  constexpr int ROWB = BR * 2;
  const int q4 = (lane & 15) >> 2, b = (lane >> 4) & 1, e = lane & 3;
  const int krow = ks * 16 + (lane >> 5) * 8 + q4;
  const unsigned char* p = tile + krow * ROWB + (((rbase >> 5) ^ q4) << 6) + 32 * b + 8 * e;
  union { bf16x4_t q[2]; bf16x8 v; } x;
  x.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p));
  x.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + 4 * ROWB));
  h = x.v;
}

template <int TM, int TN, int WR, int WC, int NBUF, int BLO = 1>
__global__ __launch_bounds__(64 * WR * WC, 2) void gemm_tn_planes_kernel(GemmArgs g) {
  constexpr int NW = WR * WC, BKT = 32;
  constexpr int BM_ = 32 * TM * WR, BN_ = 32 * TN * WC;
  static_assert(BM_ == 128 || BM_ == 256, "row pitch of the LDS image"); static_assert(BN_ == 128 || BN_ == 256, "row pitch");
  constexpr int PA = BM_ / 16, PB = BN_ / 16;                // 1 KB pieces per plane and chunk
  constexpr int NPIECE = 2 * PA + (1 + BLO) * PB, PPW = NPIECE / NW;
  static_assert(NPIECE % NW == 0, "pieces must divide evenly over the waves");
  constexpr int SLOT_B = NPIECE * 1024;                      // [A hi | A lo | B hi | B lo]
  constexpr int PATCH_B = NW * EPI_WAVE_FLOATS(TM, TN) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * SLOT_B > PATCH_B ? NBUF * SLOT_B : PATCH_B];
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wr = wave / WC, wc = wave % WC;
  // Workgroup id -> (split z, m tile, n tile). The tiles of one split that share a row panel of the LARGER operand (A when
  // M >= N: the n tiles of one m tile; else B) form a group; a group runs on ONE XCD (id % 8), its members back to back, so that
  // panel is fetched into one L2 once. The host sizes the split count so that no XCD gets more workgroups than it has CUs.
  const bool by_m = g.M >= g.N;
  const int gs = by_m ? g.ntiles : g.mtiles, og = by_m ? g.mtiles : g.ntiles;
  const int bid = (int)blockIdx.x;
  const int q = bid >> 3, G = (bid & 7) + 8 * (q / gs), mem = q % gs;
  if (G >= g.splits * og) return;
  const int z = G / og, o_i = G % og;
  const int mt_i = by_m ? o_i : mem, nt_i = by_m ? mem : o_i;
  const int64_t m0 = (int64_t)mt_i * BM_, n0 = (int64_t)nt_i * BN_;
  const int64_t kbeg = (int64_t)z * g.k_chunk;
  const int64_t kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;
  const int C = (int)((kend - kbeg) / BKT);

  // per-lane DMA sources: piece p = wave + NW * it; which plane a piece reads is wave-uniform
  uint32_t soff[PPW];
#pragma unroll
  for (int it = 0; it < PPW; ++it) {
    const int p = wave + NW * it;
    const bool isA = p < 2 * PA;
    const int pl = isA ? p % PA : (p - 2 * PA) % PB;          // piece inside its plane
    const int U = (isA ? BM_ : BN_) / 8;                      // 16-byte units per k row
    const int idx = pl * 64 + lane, krow = idx / U, pos = idx % U;
    const int unit = ((((pos >> 2) ^ (krow & 3)) << 2) | (pos & 3));
    const int64_t ld = isA ? g.lda : g.ldb, c0 = isA ? m0 : n0;
    soff[it] = (uint32_t)((krow * ld + c0 + unit * 8) * 2);
  }
  auto dma = [&](int buf, int64_t k0) {
#pragma unroll
    for (int it = 0; it < PPW; ++it) {
      const int p = wave + NW * it;                           // uniform
      const char* base = reinterpret_cast<const char*>(p < PA ? g.epi.a_hi : (p < 2 * PA ? g.epi.a_lo : (p < 2 * PA + PB ? g.epi.b_hi : g.epi.b_lo)));
      base += k0 * (p < 2 * PA ? g.lda : g.ldb) * 2;
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)(base + soff[it]), (LDS_AS void*)(smem + buf * SLOT_B + p * 1024), 16, 0, 0);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  if (C > 0) dma(0, kbeg);
  if (NBUF == 3 && C > 1) dma(1, kbeg + BKT);
  int cur = 0;
  for (int c = 0; c < C; ++c) {
    if (NBUF == 3 && c + 1 < C) wait_vmcnt<PPW>(); else wait_vmcnt<0>();   // this wave's pieces of chunk c (a younger chunk may still fly)
    __builtin_amdgcn_s_barrier();                              // everyone's pieces have landed; everyone is done with the slot refilled next
    asm volatile("" ::: "memory");
    {
      const int pc = c + NBUF - 1;
      const int pbuf = cur == 0 ? NBUF - 1 : cur - 1;
      if (pc < C) dma(pbuf, kbeg + (int64_t)pc * BKT);
    }
    const unsigned char* cA = smem + cur * SLOT_B;
    const unsigned char* cB = cA + 2 * PA * 1024;
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) read_frag_tn<BM_>(cA, wr * 32 * TM + a * 32, ks, lane, ah[a], al[a]);
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        if constexpr (BLO) read_frag_tn<BN_>(cB, wc * 32 * TN + b * 32, ks, lane, bh[b], bl[b]);
        else read_frag_tn_hi<BN_>(cB, wc * 32 * TN + b * 32, ks, lane, bh[b]);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
          if constexpr (BLO) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
    cur = cur == NBUF - 1 ? 0 : cur + 1;
  }
  gemm_epilogue<TM, TN, WR, WC, true>(g, acc, reinterpret_cast<float*>(smem), wave, lane, wr, wc, m0, n0, z, nt_i);
}


// host side of the launch (called by advmil_gemm_f32_tiled, gemm_f32.hip, which has checked the arguments)
int advmil_launch_tn_planes(int tile, bool b_single, dim3 tgrid, hipStream_t stream, const GemmArgs& g) {
  if (b_single) {
    switch (tile) {
      case 91: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 2, 4, 3, 0>), tgrid, dim3(512), 0, stream, g); break;
      case 92: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 4, 2, 3, 0>), tgrid, dim3(512), 0, stream, g); break;
      default: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 4, 4, 2, 2, 0>), tgrid, dim3(512), 0, stream, g); break;
    }
  } else {
    switch (tile) {
      case 91: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 2, 4, 3>), tgrid, dim3(512), 0, stream, g); break;   // 3 x 48 KB
      case 92: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 4, 2, 3>), tgrid, dim3(512), 0, stream, g); break;   // 3 x 48 KB
      default: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 4, 4, 2, 2>), tgrid, dim3(512), 0, stream, g); break;   // 2 x 64 KB
    }
  }
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
