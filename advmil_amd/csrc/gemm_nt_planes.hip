#include "gemm_core.h"

// =====================================================================================
// NT contraction over operands that ALREADY live in HBM as bf16 planes (hi, lo): C = epi(A B^T), A[M,K], B[N,K], both k-contiguous.
// This is the form of every forward layer applied to a step slab (embedding FCs, gate branches: A = the slab's rows or its hidden
// rows, B = a weight matrix). With both operands pre-split there is nothing to convert, so the tile goes global -> LDS by the
// gfx950 LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, no VGPR round trip, no VALU), the inner loop is ds_read_b128 + MFMA
// only, and the next chunk's DMA flies under the current chunk's MFMAs.
//   workgroup = WR x 2 waves, each wave a 64 x (32*TN) accumulator block -> tile (64*WR) x (64*TN):
//     WR = 4 (8 waves, one workgroup per CU): 256x128 / 256x192, k chunks of 32 -- the instantiated forms (256x256 measured equal);
//     WR = 2 (4 waves, two workgroups per CU, whose prologue / epilogue would hide under the other's K loop): 128x128 (k 32)
//       measured equal, 128x192 / 128x256 (k 16, 32-byte DMA rows) 10-15 % slower than the 8-wave forms on every slab shape, so
//       they are not built (tools/gemm_planes_check.py history in DESIGN.md);
//   same accumulation order as gemm_f32_kernel's bf16x3 loop -> bit-identical results;
//   LDS: NBUF buffers x [A hi | A lo | B hi | B lo] planes of [row][BKT] bf16, 16-byte units XOR-swizzled (ps_unit):
//     the DMA writes lane-linear (piece base + 16 * lane), so the swizzle is applied to each lane's SOURCE address: the lane that
//     fills stored position q of row r fetches unit ps_unit(r, q) of that row (same involution as the fragment reads);
//   one piece = 1 KB = one wave-instruction = 16 rows x 64 B (k 32) or 32 rows x 32 B (k 16).
// Requirements (checked by the host): M % (64*WR) == 0, N % (64*TN) == 0, K % BKT == 0, planes 16-byte aligned with ld % 8 == 0,
// splits == 1.
// =====================================================================================

// PERSISTENT: the grid is one workgroup per CU and each walks the tiles bid, bid + grid, ... (the same XCD-aware order as before:
// the 256 workgroups in flight at any time hold consecutive tile ids). The k-chunk ring runs ACROSS tiles: the last NBUF-1 loop
// iterations of a tile already fetch the next tile's first chunks (same per-lane source pointers plus a uniform row delta), the
// epilogue works in the ring slot of the chunk just consumed, and the next tile's first wait is a counted one that leaves the
// epilogue's stores in flight. Per tile this removes the cold start (first-chunk latency + workgroup launch: 5 of 55 us on the gate
// contraction, tools/probe/stamp_gemm.sh) and lets the C stores drain under the next tile's K loop instead of at workgroup exit.
// EPI = 1: instantiated for the fused gate score alone (its epilogue stores one partial per row and column block, no C): the
// 256x256 form, which beside the full streaming epilogue would not fit the register file. EPI = 2: 256x256 with the PLAIN streaming form.
// ALO = 0: A is a single-plane (bf16) operand -- no A lo rows in the ring (a 256x256 chunk stages 48 KB instead of 64), two MFMAs
// per product (a.b = ah.bh + ah.bl exactly as the three-product form with al = 0: same order, bit-identical to it).
template <int TN, int NBUF, int WR, int BKT, int EPI = 0, int ALO = 1>
__global__ __launch_bounds__(128 * WR, 2) void gemm_nt_planes_kernel(GemmArgs g) {
  constexpr int TM = 2, WC = 2, NW = WR * WC;
  constexpr int BM_ = 64 * WR, BN_ = 64 * TN;
  constexpr int AROWS = (1 + ALO) * BM_;               // plane rows of A per buffer
  constexpr int ROWS_ALL = AROWS + 2 * BN_;            // plane rows per buffer: A hi, (A lo,) B hi, B lo
  constexpr int RPP = 512 / BKT, LPR = BKT / 8;        // rows per 1 KB piece, lanes (16-byte units) per row
  constexpr int NPIECE = ROWS_ALL / RPP, PPW = NPIECE / NW;
  static_assert(NPIECE % NW == 0, "pieces must divide evenly over the waves");
  constexpr int BUF_HW = ROWS_ALL * BKT;               // halfwords per buffer
  constexpr int PATCH_FLOATS = NW * EPI_WAVE_FLOATS(TM, TN);
  // the epilogue's LDS area: the ring slot of the chunk just consumed when it fits one (every two-plane-A form), else an area of its
  // own behind the ring (the narrower single-plane-A slots of the 256x128 / 256x192 tiles)
  constexpr bool EPI_IN_SLOT = PATCH_FLOATS <= BUF_HW / 2;
  static_assert(EPI_IN_SLOT || ALO == 0, "the epilogue area must fit one ring slot");
  constexpr int EST = TM * TN * 4;                     // stores per wave of the streaming epilogue (its lower bound)
  static_assert(EST + PPW <= 63, "counted waits are 6-bit");
  __shared__ __attribute__((aligned(16))) float smem[NBUF * BUF_HW / 2 + (EPI_IN_SLOT ? 0 : PATCH_FLOATS)];
  bf16raw* const lds = reinterpret_cast<bf16raw*>(smem);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // uniform: lives in an SGPR
  const int wr = wave / WC, wc = wave % WC;
  const int ntile = g.mtiles * g.ntiles, G = (int)gridDim.x;
  auto tile_of = [&](int v, int& mt, int& nt) {   // XCD-aware tile order (see gemm_f32_kernel): the n-tiles of one A row panel run on one XCD
    const int inner = g.ntiles, outer = g.mtiles;
    const int per_group = 8 * inner, full = (outer / 8) * per_group;
    if (v < full) {
      const int r = v % per_group;
      mt = (v / per_group) * 8 + (r & 7);
      nt = r >> 3;
    } else {
      const int rem = outer - (outer / 8) * 8, r = v - full;
      mt = (outer / 8) * 8 + r % rem;
      nt = r / rem;
    }
  };
  int v = (int)blockIdx.x;
  int mt_i, nt_i;
  tile_of(v, mt_i, nt_i);
  // per-lane DMA sources: piece p = wave + NW*it covers plane rows [RPP*p, RPP*p + RPP) of the buffer image; which operand a piece
  // belongs to is wave-uniform
  // per-lane DMA sources: piece p = wave + NW*it covers plane rows [RPP*p, RPP*p + RPP) of the buffer image. Which plane a piece
  // reads is wave-uniform, so a lane keeps one 32-bit byte offset per piece (planes < 4 GB: host check) and the plane's base, the
  // k offset and the tile-to-tile row delta are added on the scalar side.
  uint32_t soff[PPW];
  // (recomputed for every tile instead of carried across the epilogue: there the accumulators + the streaming epilogue's operands
  // already fill the register file)
  auto set_src = [&](int mt, int nt, int lane) {
#pragma unroll
    for (int it = 0; it < PPW; ++it) {
      const int prow = (wave + NW * it) * RPP + lane / LPR;  // row in the buffer image
      int r;                                                 // tile-local row of its operand
      int64_t ld, row0;
      if (prow < BM_) { r = prow; ld = g.lda; row0 = (int64_t)mt * BM_; }
      else if (prow < AROWS) { r = prow - BM_; ld = g.lda; row0 = (int64_t)mt * BM_; }
      else if (prow < AROWS + BN_) { r = prow - AROWS; ld = g.ldb; row0 = (int64_t)nt * BN_; }
      else { r = prow - AROWS - BN_; ld = g.ldb; row0 = (int64_t)nt * BN_; }
      soff[it] = (uint32_t)(((row0 + r) * ld + ps_unit(r, lane % LPR, LPR) * 8) * 2);
    }
  };
  set_src(mt_i, nt_i, (int)threadIdx.x & 63);
  int64_t dA = 0, dB = 0;                                  // element offsets from this tile's rows to the next tile's
  auto dma = [&](int buf, int64_t k0, bool next) {
#pragma unroll
    for (int it = 0; it < PPW; ++it) {
      const int p0 = (wave + NW * it) * RPP;               // uniform
      const char* base = reinterpret_cast<const char*>(p0 < BM_ ? g.epi.a_hi : (p0 < AROWS ? g.epi.a_lo : (p0 < AROWS + BN_ ? g.epi.b_hi : g.epi.b_lo)));
      base += (k0 + (next ? (p0 < AROWS ? dA : dB) : 0)) * 2;
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)(base + soff[it]), (LDS_AS void*)(lds + buf * BUF_HW + (wave + NW * it) * 512), 16, 0, 0);
    }
  };
  const int64_t K = g.K;
  const int C = (int)(K / BKT);                            // >= NBUF - 1 (host check)
  dma(0, 0, false);
  if (NBUF == 3) dma(1, BKT, false);
  int cur = 0;
  int young = 0;                                           // stores of the previous tile's epilogue still allowed in flight
  for (; v < ntile; v += G) {
    // Every per-lane constant of a tile (fragment addresses, epilogue geometry) is re-derived from an opaque copy of the lane id, so
    // none of them is carried in a register across the epilogue of the previous tile (where the file is full: carried, they spilled)
    int lane = (int)threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int i = lane & 31, hi = lane >> 5;
    const bool has_next = v + G < ntile;
    const int64_t m0 = (int64_t)mt_i * BM_, n0 = (int64_t)nt_i * BN_;
    int mt_n = mt_i, nt_n = nt_i;
    if (has_next) tile_of(v + G, mt_n, nt_n);
    dA = (int64_t)(mt_n - mt_i) * BM_ * g.lda;
    dB = (int64_t)(nt_n - nt_i) * BN_ * g.ldb;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    for (int c = 0; c < C; ++c) {
      // counted wait: this wave's pieces of chunk c have landed; what may stay in flight is everything issued after them -- the
      // younger chunk's PPW pieces (three buffers) and, in the first NBUF-1 iterations of a tile, the previous epilogue's stores
      // (a raw s_barrier: __syncthreads would drain vmcnt to 0 because an LDS-DMA is a pending LDS write)
      const bool more = c + 1 < C || has_next;             // a younger chunk was issued (three buffers)
      if constexpr (NBUF == 3) {
        const bool st = young && c < 2;
        if (more) { if (st) wait_vmcnt<PPW + EST>(); else wait_vmcnt<PPW>(); }
        else { if (st) wait_vmcnt<EST>(); else wait_vmcnt<0>(); }
      } else {
        if (young && c == 0) wait_vmcnt<EST>(); else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();                        // everyone's pieces have; and everyone is done with the slot refilled next
      asm volatile("" ::: "memory");
      {
        const int pc = c + NBUF - 1;                       // chunk to prefetch, into the slot of chunk c - 1
        const int pbuf = cur == 0 ? NBUF - 1 : cur - 1;
        if (pc < C) dma(pbuf, (int64_t)pc * BKT, false);
        else if (has_next) dma(pbuf, (int64_t)(pc - C) * BKT, true);
      }
      const bf16raw* cA = lds + cur * BUF_HW;
      const bf16raw* cB = cA + AROWS * BKT;
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          if constexpr (ALO) read_frag_presplit<BM_, BKT>(cA, wr * 32 * TM + a * 32, ks, i, hi, ah[a], al[a]);
          else read_frag_hi<BKT>(cA, wr * 32 * TM + a * 32, ks, i, hi, ah[a]);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) read_frag_presplit<BN_, BKT>(cB, wc * 32 * TN + b * 32, ks, i, hi, bh[b], bl[b]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            if constexpr (ALO) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
          }
      }
      cur = cur == NBUF - 1 ? 0 : cur + 1;
    }
    // the epilogue's LDS area: the ring slot of the chunk just consumed (the other slots hold / receive the next tile's chunks; the
    // slot is refilled only behind the next tile's first barrier, which every wave reaches after it has left the epilogue)
    const int last = cur == 0 ? NBUF - 1 : cur - 1;
    young = gemm_epilogue<TM, TN, WR, WC, true, EPI>(g, acc, EPI_IN_SLOT ? smem + last * (BUF_HW / 2) : smem + NBUF * (BUF_HW / 2), wave, lane, wr,
                                                     wc, m0, n0, 0, nt_i);
    mt_i = mt_n; nt_i = nt_n;
    if (has_next) {
      int lane2 = (int)threadIdx.x & 63;
      asm volatile("" : "+v"(lane2));
      set_src(mt_i, nt_i, lane2);
    }
  }
}


// host side of the launch (called by advmil_gemm_f32_tiled, gemm_f32.hip, which has checked the arguments)
int advmil_launch_nt_planes(int tile, bool a_single, dim3 pgrid, hipStream_t stream, const GemmArgs& g) {
  if (a_single) {                                       // A = a bf16 slab (x_storage = "bf16"): two products per MFMA step
    switch (tile) {
      case 85: hipLaunchKernelGGL((gemm_nt_planes_kernel<4, 2, 4, 32, 2, 0>), pgrid, dim3(512), 0, stream, g); break;
      case 86: hipLaunchKernelGGL((gemm_nt_planes_kernel<2, 3, 4, 32, 2, 0>), pgrid, dim3(512), 0, stream, g); break;
      case 83: hipLaunchKernelGGL((gemm_nt_planes_kernel<3, 2, 4, 32, 0, 0>), pgrid, dim3(512), 0, stream, g); break;
      case 82: hipLaunchKernelGGL((gemm_nt_planes_kernel<2, 3, 4, 32, 0, 0>), pgrid, dim3(512), 0, stream, g); break;
      default: return ADVMIL_EINVAL;
    }
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  switch (tile) {
    case 84: hipLaunchKernelGGL((gemm_nt_planes_kernel<4, 2, 4, 32, 1>), pgrid, dim3(512), 0, stream, g); break;   // 2 x 64 KB
    case 85: hipLaunchKernelGGL((gemm_nt_planes_kernel<4, 2, 4, 32, 2>), pgrid, dim3(512), 0, stream, g); break;   // 256x256, plain streaming epilogue (+ two layers)
    case 86: hipLaunchKernelGGL((gemm_nt_planes_kernel<2, 3, 4, 32, 2>), pgrid, dim3(512), 0, stream, g); break;   // 256x128, plain streaming epilogue (+ two layers)
    case 83: hipLaunchKernelGGL((gemm_nt_planes_kernel<3, 2, 4, 32>), pgrid, dim3(512), 0, stream, g); break;   // 2 x 56 KB
    case 82: hipLaunchKernelGGL((gemm_nt_planes_kernel<2, 3, 4, 32>), pgrid, dim3(512), 0, stream, g); break;   // 3 x 48 KB: two chunks in flight
    default: return ADVMIL_EINVAL;
  }
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
