// Host side of the contraction engine: arithmetic mode, launch plans, argument checks and the dispatch to the kernels' translation
// units (gemm_f32.hip: generic kernels, grouped launch, split-K reduce; gemm_nt_planes.hip / gemm_tn_planes.hip: the plane-fed kernels).
#include "gemm_core.h"

int advmil_launch_generic(int tile, int a_kc, int b_kc, int pre, dim3 grid, hipStream_t stream, const GemmArgs& g);
int advmil_launch_nt_planes(int tile, bool a_single, dim3 pgrid, hipStream_t stream, const GemmArgs& g);
int advmil_launch_tn_planes(int tile, bool b_single, dim3 tgrid, hipStream_t stream, const GemmArgs& g);
int advmil_launch_splitk_reduce(const GemmArgs& g, hipStream_t stream);

extern "C" size_t advmil_gemm_f32_workspace_bytes(int64_t M, int64_t N, int splits) {
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-bf16 ("bf16x3") on the bf16 matrix pipe
int g_gemm_mode = 0;             // (read by the kernels' launchers in gemm_f32.hip)
static int g_nt_planes = []() { const char* e = getenv("ADVMIL_NT_PLANES"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" int advmil_set_gemm_mode(int mode) {
  if (mode != 0 && mode != 1) return ADVMIL_EINVAL;
  g_gemm_mode = mode;
  return ADVMIL_OK;
}
extern "C" int advmil_get_gemm_mode(void) { return g_gemm_mode; }

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
  if (tile == 85 || tile == 86) return (int)(N / (64 * (tile == 85 ? 4 : 2))) * 2;   // (its plain forms: the training gate score)
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
    case 83: case 82: tm = 2; tn = tile % 10; wr = 4; wc = 2; return true;      // plane-fed NT kernel, full epilogue (256x192 / 256x128)
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
  // (... and of LAYER 1 of a two-layer launch, tiles 85 / 86: checked with the tile below)
  if (!C && !epi->gate_wc && (epi->accumulate || splits != 1 || (epi->c2 && tile != 85 && tile != 86))) return ADVMIL_EINVAL;
  if ((lda & 3) || (ldb & 3)) return ADVMIL_EINVAL;
  if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N)) return ADVMIL_EINVAL;          // a row pitch shorter than the row it strides
  if (ldc < (epi->c2 ? (int64_t)epi->n_split : N) && (C || epi->c_hi)) return ADVMIL_EINVAL;      // (two-layer form: C holds the first n_split columns)
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
  if (epi->c_rows_pair32 && (splits < 2 || !epi->accumulate || (M & 63) || ldc != N)) return ADVMIL_EINVAL;
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
  const bool gate_store = epi->gate_wc && epi->gate_bits_a;      // training form (tiles 85 / 86): C is written, dropout through given keep bits
  if (gate_store && (!epi->gate_bits_b || !C || (tile != 85 && tile != 86) || epi->seed || !(epi->drop_p > 0.0f) || epi->drop_p >= 1.0f ||
                     (N & 127) || epi->ldgbits < N / 64 || epi->c2 || epi->c_hi || (((uintptr_t)epi->gate_wc) & 15)))
    return ADVMIL_EINVAL;
  if (epi->gate_wc) {       // fused gate score: no split-K, no dropout, whole float4 column groups, one partial per 32*TN*... block
    if (splits != 1 || !epi->gate_out || (N & 3) || (epi->drop_p > 0.0f && !gate_store)) return ADVMIL_EINVAL;
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
    if (plain && ((epi->gate_wc && !gate_store) || epi->rowv || epi->maskref || epi->accumulate || (epi->seed && epi->drop_p > 0.0f))) return ADVMIL_EINVAL;
    if (epi->c2) {      // two layers in one launch: the plain forms only, split on a 32-column boundary inside N
      if (!plain || epi->n_split <= 0 || epi->n_split >= N || (epi->n_split & 31) || (epi->ldc2 & 3) || ((uintptr_t)epi->c2 & 15) ||
          epi->act_split != epi->n_split)
        return ADVMIL_EINVAL;
    }
    if (g_gemm_mode != 1 || !a_kc || !b_kc || pre != 3 || splits != 1 || (M % bm) || (K % bkt) || (N % (64 * tnp))) return ADVMIL_EINVAL;
    if (epi->gate_wc && (!epi->gate_out || (epi->drop_p > 0.0f && !gate_store) || epi->gate_np != advmil_gemm_f32_gate_blocks(tile, N))) return ADVMIL_EINVAL;
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
      const int rc = advmil_launch_splitk_reduce(g, stream);
      if (rc) return rc;
    }
    return ADVMIL_OK;
  }
  { const int rc = advmil_launch_generic(tile, a_kc, b_kc, pre, grid, stream, g); if (rc) return rc; }
  if (splits > 1) {
    const int rc = advmil_launch_splitk_reduce(g, stream);
    if (rc) return rc;
  }
  return ADVMIL_OK;
}

extern "C" int advmil_gemm_f32(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                               const float* B, int64_t ldb, float* C, int64_t ldc, const advmil_epilogue_t* epi,
                               int splits, void* ws, size_t ws_bytes, advmil_stream_t stream) {
  return advmil_gemm_f32_tiled(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, epi, splits, 0, ws, ws_bytes, stream);
}
