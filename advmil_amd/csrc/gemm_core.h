// Shared device code of the contraction engine (gemm_f32.hip: generic kernels + host dispatch; gemm_nt_planes.hip / gemm_tn_planes.hip:
// the plane-fed LDS-DMA kernels): launch arguments, operand staging, fragment reads and the epilogues. Three translation units so that
// they compile side by side (one file took five minutes).
#pragma once
#include <cstdlib>
#include "common.h"
#include "bf16split.h"
#include "sumq.h"
#include "../../include/advmil_hip.h"

#define BK 32          // k per chunk of the exact-fp32 variant (and the granularity of split-K chunking)
#define PITCH_KC 36    // BK + 4

// LDS ordering inside ONE wave (a wave's DS operations complete in issue order; this only stops the compiler from moving them
// across and drains the queue): enough when the LDS region is private to the wave.
#define WAVE_LDS_SYNC()                                   \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_s_waitcnt(0xc07f);                   \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

struct GemmArgs {
  int64_t M, N, K;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int64_t k_chunk;   // K range per split (multiple of BK)
  float* ws;         // [splits][M][N] partials when splits > 1
  int splits;
  int mtiles, ntiles;
  advmil_epilogue_t epi;
};

__device__ __forceinline__ float epilogue_elem(const advmil_epilogue_t& e, float acc, int64_t m, int64_t n, int64_t N,
                                               uint64_t key, float inv_keep) {
  float v = acc * e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.rowv) v += e.rowv[m] * e.colv[(e.rowseg ? (int64_t)e.rowseg[m] * N : 0) + n];
  v = act_apply(n < e.act_split ? e.act0 : e.act1, v);
  if (e.seed && e.drop_p > 0.0f) v *= rng_keep(key, (uint64_t)((e.rng_row ? e.rng_row[m] : m) * N + n), e.drop_p, inv_keep);
  if (e.maskref) v *= (e.maskref[m * (int64_t)e.ldmask + n] > 0.0f ? e.mask_scale : 0.0f);
  return v;
}

// ROWS = 64*T rows (m or n) x 32 k per tile; P = ROWS/32 float4 per thread.
template <bool KC, int ROWS, int BKT, int NT>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t rows,
                                          int64_t k0, int64_t kend, int tid, float4 (&r)[ROWS * BKT / (4 * NT)]) {
#pragma unroll
  for (int p = 0; p < ROWS * BKT / (4 * NT); ++p) {
    const int e = p * NT + tid;
    if (KC) {   // [row][k]: BKT/4 float4 per row
      const int64_t row = row0 + e / (BKT / 4);
      const int64_t k = k0 + (e % (BKT / 4)) * 4;
      r[p] = (row < rows && k < kend) ? *reinterpret_cast<const float4*>(src + row * ld + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {    // [k][m]: ROWS/4 float4 per k
      const int64_t k = k0 + e / (ROWS / 4);
      const int64_t m = row0 + (e % (ROWS / 4)) * 4;
      r[p] = (k < kend && m < rows) ? *reinterpret_cast<const float4*>(src + k * ld + m) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <bool KC, int ROWS, int BKT, int NT>
__device__ __forceinline__ void store_tile(float* __restrict__ s, int tid, const float4 (&r)[ROWS * BKT / (4 * NT)]) {
#pragma unroll
  for (int p = 0; p < ROWS * BKT / (4 * NT); ++p) {
    const int e = p * NT + tid;
    if (KC)
      *reinterpret_cast<float4*>(s + (e / (BKT / 4)) * (BKT + 4) + (e % (BKT / 4)) * 4) = r[p];
    else
      *reinterpret_cast<float4*>(s + (e / (ROWS / 4)) * ROWS + (e % (ROWS / 4)) * 4) = r[p];
  }
}

// fragment for the 32-row block starting at `rbase`, k-group t4: f[u], u = 0..3
template <bool KC, int ROWS, int BKT>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int rbase, int t4, int i, int hi, float (&f)[4]) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4*>(s + (rbase + i) * (BKT + 4) + t4 * 8 + hi * 4);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = s[(t4 * 8 + hi * 4 + u) * ROWS + rbase + i];
  }
}

// ---- split-bf16 ("bf16x3") arithmetic: x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a.b ~= ah.bh + ah.bl + al.bh on the
// bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate), fp32 accumulate. The dropped terms are ~2^-17 of |a||b|
// per product (fp32 MFMA: 2^-24), i.e. near-fp32 results at 3/16 of the matrix-pipe time. Operands stay fp32 in HBM; the split
// happens once per staged element on the way into LDS, which holds two bf16 planes (hi, lo) per operand tile.

// Pre-split LDS image of a k-contiguous operand tile (bf16x3 mode): two bf16 planes [row][BKT], hi then lo, written once per
// element when the tile is staged (every element is consumed by two waves, so splitting here halves the conversion work and
// leaves the inner loop with ds_read_b128 + MFMA only). Rows are NOT padded; instead the 16-byte unit u (8 consecutive k) of row r
// lives at unit u ^ ((r >> 2) & (BKT/8 - 1)): the b128 fragment reads of a 16-lane group ({0-3,12-15,20-27} and their shifts)
// then cover all 64 banks, and the 8-byte staging stores of a 16-lane group cover two whole rows = 32 distinct banks. (A padded
// 80-byte pitch measured 33% of LDS cycles as bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r01_pmc_sq_bf16x3.json.)
#define PITCH_PS(BKT) (BKT)
// (two units per row, k chunks of 16: rows of 32 B -> the 16 lanes of a b128 group {0-3,12-15,20-27} cover all 64 banks when the
// halves swap every 8 rows, not every 4)
__device__ __forceinline__ int ps_unit(int row, int unit, int units_per_row) {
  return units_per_row == 2 ? (unit ^ ((row >> 3) & 1)) : (unit ^ ((row >> 2) & (units_per_row - 1)));
}

template <int ROWS, int BKT, int NT>
__device__ __forceinline__ void store_tile_presplit(bf16raw* __restrict__ planes, int tid, const float4 (&r)[ROWS * BKT / (4 * NT)]) {
  bf16raw* const hiP = planes;
  bf16raw* const loP = planes + ROWS * PITCH_PS(BKT);
#pragma unroll
  for (int p = 0; p < ROWS * BKT / (4 * NT); ++p) {
    const int e = p * NT + tid;
    const int row = e / (BKT / 4), p4 = e % (BKT / 4);        // p4: which 4-k piece of the row
    const int off = row * PITCH_PS(BKT) + ps_unit(row, p4 >> 1, BKT / 8) * 8 + (p4 & 1) * 4;
    uint2 h, l;
    split4(r[p], h, l);
    *reinterpret_cast<uint2*>(hiP + off) = h;
    *reinterpret_cast<uint2*>(loP + off) = l;
  }
}

template <int ROWS, int BKT>
__device__ __forceinline__ void read_frag_presplit(const bf16raw* __restrict__ planes, int rbase, int ks, int i, int hi,
                                                   bf16x8& h, bf16x8& l) {
  const int row = rbase + i;
  const bf16raw* p = planes + row * PITCH_PS(BKT) + ps_unit(row, ks * 2 + hi, BKT / 8) * 8;
  Frag8 a, b;
  a.u = *reinterpret_cast<const uint4*>(p);
  b.u = *reinterpret_cast<const uint4*>(p + ROWS * PITCH_PS(BKT));
  h = a.v; l = b.v;
}

// the hi plane's fragment alone (single-plane operands)
template <int BKT>
__device__ __forceinline__ void read_frag_hi(const bf16raw* __restrict__ planes, int rbase, int ks, int i, int hi, bf16x8& h) {
  const int row = rbase + i;
  Frag8 a;
  a.u = *reinterpret_cast<const uint4*>(planes + row * PITCH_PS(BKT) + ps_unit(row, ks * 2 + hi, BKT / 8) * 8);
  h = a.v;
}

// Pre-split LDS image of an m-contiguous operand tile ([k][rows] source, e.g. both operands of dW = dY^T X): two bf16 planes
// [k][PITCH_MC(ROWS)], hi then lo, in the SOURCE orientation (coalesced 8-byte stores), consumed with the gfx950 LDS transpose
// read: ds_read_b64_tr_b16 hands lane c of a 16-lane group the 4 k-consecutive halfwords of column c out of a [4 k][16 m]
// block whose 16 8-byte pieces are addressed by the group's lanes (lane s: k row s>>2, m piece s&3; measured with
// tools/probe/tr_probe.hip). Two such reads give a lane the 8 consecutive k of its row that v_mfma_f32_32x32x16_bf16 wants, with no
// per-element LDS traffic. Pitch = ROWS*2 + 64 bytes: the 4 k-rows of one read land in 4 different 64-byte bank quarters.
#define PITCH_MC(ROWS) ((ROWS) + 32)

template <int ROWS, int BKT, int NT>
__device__ __forceinline__ void store_tile_presplit_mc(bf16raw* __restrict__ planes, int tid, const float4 (&r)[ROWS * BKT / (4 * NT)]) {
  bf16raw* const hiP = planes;
  bf16raw* const loP = planes + BKT * PITCH_MC(ROWS);
#pragma unroll
  for (int p = 0; p < ROWS * BKT / (4 * NT); ++p) {
    const int e = p * NT + tid;
    const int k = e / (ROWS / 4), m4 = (e % (ROWS / 4)) * 4;
    uint2 h, l;
    split4(r[p], h, l);
    *reinterpret_cast<uint2*>(hiP + k * PITCH_MC(ROWS) + m4) = h;
    *reinterpret_cast<uint2*>(loP + k * PITCH_MC(ROWS) + m4) = l;
  }
}

template <int ROWS, int BKT>
__device__ __forceinline__ void read_frag_presplit_mc(const bf16raw* __restrict__ planes, int rbase, int ks, int lane,
                                                      bf16x8& h, bf16x8& l) {
  // lane -> (k row of its 8-byte piece, m piece); the result lane holds row rbase + (lane & 31), k = ks*16 + (lane>>5)*8 + 0..7
  const int krow = ks * 16 + (lane >> 5) * 8 + ((lane & 15) >> 2);
  const int mcol = rbase + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  const bf16raw* p = planes + krow * PITCH_MC(ROWS) + mcol;
  union { bf16x4_t q[2]; bf16x8 v; } a, b;
  a.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p));
  a.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + 4 * PITCH_MC(ROWS)));
  b.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + BKT * PITCH_MC(ROWS)));
  b.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + (BKT + 4) * PITCH_MC(ROWS)));
  h = a.v; l = b.v;
}

// planes of 4 consecutive final outputs (row pitch = ldc): what the next contraction consumes without re-splitting
__device__ __forceinline__ void emit_planes4(const advmil_epilogue_t& e, int64_t off, const float (&v)[4], int nvalid) {
  union { __bf16 b[4]; uint2 u; bf16raw r[4]; } h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h.b[j] = (__bf16)v[j];
    l.b[j] = (__bf16)(v[j] - (float)h.b[j]);
  }
  bf16raw* ch = reinterpret_cast<bf16raw*>(e.c_hi) + off;
  bf16raw* cl = reinterpret_cast<bf16raw*>(e.c_lo) + off;
  if (nvalid == 4 && ((off & 3) == 0)) {
    *reinterpret_cast<uint2*>(ch) = h.u;
    *reinterpret_cast<uint2*>(cl) = l.u;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) { ch[j] = h.r[j]; cl[j] = l.r[j]; }
  }
}

// One operand's register stage in bf16x3 mode: global -> registers (load) -> the pre-split LDS image (store).
// PRE = false: the source is the fp32 matrix; the hi/lo split happens in store() (once per staged element).
// PRE = true : the source is a pair of bf16 planes the caller already holds in HBM (advmil_split_planes, the Adam kernel for the
//   weights, or a producing GEMM's epilogue): same bytes from memory, 16-byte pieces straight into LDS, no conversion work at
//   all. That matters because the bf16x3 loop is instruction-issue bound (PMC: ~255 VALU per 24 MFMA per k-chunk and wave, half
//   of them this split, repeated by every workgroup that re-reads the element: 3x for X, 1024x for a weight).
template <bool KC, int ROWS, int BKT, bool PRE, int NT>
struct OperandStage {
  static constexpr int NF4 = ROWS * BKT / (4 * NT);   // float4 per thread (fp32 source)
  static constexpr int NPIECE = ROWS * BKT / 8;       // 16-byte pieces per plane
  static constexpr int NP = (NPIECE + NT - 1) / NT;   // per thread (the 192-row tile on 512 threads: 1.5 -> 2, the tail predicated)
  static constexpr bool FULL = NPIECE % NT == 0;
  float4 f[PRE ? 1 : NF4];
  uint4 ph[PRE ? NP : 1], pl[PRE ? NP : 1];

  __device__ __forceinline__ void load(const float* __restrict__ src, const bf16raw* __restrict__ hi, const bf16raw* __restrict__ lo,
                                       int64_t ld, int64_t row0, int64_t rows, int64_t k0, int64_t kend, int tid) {
    if constexpr (!PRE) {
      load_tile<KC, ROWS, BKT, NT>(src, ld, row0, rows, k0, kend, tid, f);
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int e = p * NT + tid;
        int64_t off;
        bool ok;
        if (KC) {   // [row][k]: BKT/8 pieces per row
          const int64_t row = row0 + e / (BKT / 8), k = k0 + (e % (BKT / 8)) * 8;
          ok = row < rows && k < kend && (FULL || e < NPIECE);
          off = row * ld + k;
        } else {    // [k][m]: ROWS/8 pieces per k
          const int64_t k = k0 + e / (ROWS / 8), m = row0 + (e % (ROWS / 8)) * 8;
          ok = k < kend && m < rows && (FULL || e < NPIECE);
          off = k * ld + m;
        }
        ph[p] = ok ? *reinterpret_cast<const uint4*>(hi + off) : make_uint4(0u, 0u, 0u, 0u);
        pl[p] = (ok && lo) ? *reinterpret_cast<const uint4*>(lo + off) : make_uint4(0u, 0u, 0u, 0u);      // (single-plane operand: lo == NULL)
      }
    }
  }

  __device__ __forceinline__ void store(bf16raw* __restrict__ planes, int tid) const {
    if constexpr (!PRE) {
      if (KC) store_tile_presplit<ROWS, BKT, NT>(planes, tid, f);
      else store_tile_presplit_mc<ROWS, BKT, NT>(planes, tid, f);
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int e = p * NT + tid;
        if (!FULL && e >= NPIECE) break;
        if (KC) {
          const int row = e / (BKT / 8);
          bf16raw* d = planes + row * PITCH_PS(BKT) + ps_unit(row, e % (BKT / 8), BKT / 8) * 8;
          *reinterpret_cast<uint4*>(d) = ph[p];
          *reinterpret_cast<uint4*>(d + ROWS * PITCH_PS(BKT)) = pl[p];
        } else {
          bf16raw* d = planes + (e / (ROWS / 8)) * PITCH_MC(ROWS) + (e % (ROWS / 8)) * 8;
          *reinterpret_cast<uint4*>(d) = ph[p];
          *reinterpret_cast<uint4*>(d + BKT * PITCH_MC(ROWS)) = pl[p];
        }
      }
    }
  }
};

// ---- epilogue (shared by the contraction kernels). MFMA C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5). Each 32x32
// accumulator tile goes through a per-wave LDS patch ([32][36] floats, carved from the operand buffers) so that the math below runs
// once per float4 in a compact loop and the global stores are 16 B per lane along the row.
// Per-wave LDS area of the epilogue: the [32][36] accumulator patch + the streaming form's column / row side data.
#define EPI_WAVE_FLOATS(TM, TN) (32 * PITCH_KC + 32 * ((TN) + 2 * (TM)) + 32 * (TM) * (TN))      // patch | bias, row data | mask words

// Streaming form of the epilogue for launches that cover M and N with whole tiles (M % tile height == 0, N % tile width == 0, one
// activation per 32 columns, 16-byte aligned operands, at most one of the rank-1 / mask / accumulate modes).
// vmcnt counts loads AND stores in issue order on gfx950, so a global load whose result is waited for behind a store drains every
// older store first: with the bias / row vectors fetched inside the tile loop the workgroup had one sub-tile (4 KB per wave) of C
// in flight at a time and the epilogue alone ran at ~2 TB/s (403 MB of gate activations: 209 us, tools/probe/ablate_gemm.sh
// noloop). Here the store stream never waits on a load issued behind it:
//   * the wave's bias slice and its rows' data (dropout stream row, rank-1 row factor, bag index) are fetched once, before the
//     first store, and parked in the wave's LDS side area: inside the loop they come back through lgkmcnt, not vmcnt;
//   * the per-element operand of the rank-1 / mask / accumulate mode (one of colv, maskref, C) is fetched one sub-tile ahead, issued
//     after the current sub-tile's math and before its stores.
// The plain bias + activation (+ dropout, + planes) modes then have no global load inside the loop at all.
// PLAIN: instantiated for launches that are known to be bias + activation (+ planes) only (the caller checks): no dropout, no
// per-element operand -- 30 registers less, which is what lets the 256x256 persistent tile carry this epilogue.
template <int TM, int TN, int WR, int WC, bool PLAIN = false>
__device__ __forceinline__ void gemm_epilogue_stream(const GemmArgs& g, f32x16 (&acc)[TM][TN], float* patch, int lane, int wr, int wc,
                                                     int64_t m0, int64_t n0, uint64_t key, float inv_keep, int nt_i = 0) {
  const advmil_epilogue_t& e = g.epi;
  const int i = lane & 31, hi = lane >> 5;
  const int c4 = (lane & 7) * 4, rq = lane >> 3;
  const int64_t N = g.N, ldo = g.ldc;
  float* const out = g.C;
  const bool drop = !PLAIN && e.seed && e.drop_p > 0.0f;
  const bool mapped = drop && e.rng_row;
  const int kind = PLAIN ? 0 : (e.rowv ? 1 : (e.maskref ? 2 : (e.accumulate ? 3 : 0)));     // which per-element operand is fetched one sub-tile ahead
  // rank-1 term AND mask in one launch (dh = dG Wab + A dpooled, masked by the first layer's stored output: that layer's activation /
  // dropout backward rides in this epilogue instead of a row pass of its own): a second prefetched per-element operand
  const bool rmask = !PLAIN && kind == 1 && e.maskref != nullptr;
  const bool csum = !PLAIN && e.colsum != nullptr;      // per-wave column sums of the final values (the bias gradient of that layer)
  const bool bmask = !PLAIN && e.maskbits != nullptr;   // the mask as bits: the wave's words are parked in LDS, no load in the store loop
  // PLAIN launches only -- the TRAINING pass of the gated attention scorer (round 6): B's rows are the two branches in blocks of 32
  // ([a_0..31 | b_0..31 | a_32..63 | ...]: advmil_gate_interleave, pair32), so sub-tile 2u of a wave holds tanh branch columns j and
  // sub-tile 2u + 1 the sigmoid branch columns of the SAME j in the same lanes. The activations are stored as always (the backward
  // needs them); in passing, each row's  sum_j (a_j keep_a)(b_j keep_b) wc_j  over the wave's columns goes to gate_out -- the score
  // pass over the stored [rows, 2D] activations (403 MB at the 16-bag slab) never runs. The keep bits come from gate_bits_a / _b
  // (advmil_dropout_planes draws them beside the first layer's own mask): parked in LDS, no hash in the epilogue.
  const bool gpair = PLAIN && e.gate_wc != nullptr && e.gate_bits_a != nullptr;
  const float* const xbase = kind == 1 ? e.colv : (kind == 2 ? e.maskref : out);
  const int64_t xld = kind == 1 ? N : (kind == 2 ? (int64_t)e.ldmask : ldo);
  const int64_t rbase = m0 + wr * 32 * TM, cbase = n0 + wc * 32 * TN;
  const int64_t nsp = (PLAIN && e.c2) ? e.n_split : N;          // columns >= nsp belong to the launch's second layer (two-layer form)
  // ---- side data -> LDS: sbias[32*TN] | srow_i[32*TM] (bag index, or dropout stream row: never both in one launch) | srow_f[32*TM]
  float* const sbias = patch + 32 * PITCH_KC;
  int* const srow_i = reinterpret_cast<int*>(sbias + 32 * TN);
  float* const srow_f = sbias + 32 * TN + 32 * TM;
  uint32_t* const smask = reinterpret_cast<uint32_t*>(sbias + 32 * TN + 64 * TM);      // [32 TM rows][TN words]
  if (gpair) {
    // words of the wave's 32 TM rows x (TN / 2) pair blocks x 2 branches -> smask[(row * (TN / 2) + u) * 2 + branch]; wc of its 16 TN
    // branch columns -> srow_f (a plain launch has no row data there)
    constexpr int NW = 32 * TM * TN;
    uint32_t mv[(NW + 63) / 64];
#pragma unroll
    for (int u = 0; u < (NW + 63) / 64; ++u) {
      const int idx = u * 64 + lane;
      const int br = idx & 1, pu = (idx >> 1) % (TN / 2), row = (idx >> 1) / (TN / 2);
      const uint32_t* src = br ? e.gate_bits_b : e.gate_bits_a;
      mv[u] = idx < NW ? src[(rbase + row) * e.ldgbits + (cbase >> 6) + pu] : 0u;
    }
#pragma unroll
    for (int u = 0; u < (NW + 63) / 64; ++u)
      if (u * 64 + lane < NW) smask[u * 64 + lane] = mv[u];
  }
  if (bmask) {
    uint32_t mv[(32 * TM * TN + 63) / 64];
#pragma unroll
    for (int u = 0; u < (32 * TM * TN + 63) / 64; ++u) {
      const int idx = u * 64 + lane;
      mv[u] = idx < 32 * TM * TN ? e.maskbits[(rbase + idx / TN) * e.ldbits + (cbase >> 5) + idx % TN] : 0u;
    }
#pragma unroll
    for (int u = 0; u < (32 * TM * TN + 63) / 64; ++u)
      if (u * 64 + lane < 32 * TM * TN) smask[u * 64 + lane] = mv[u];
  }
  {
    float bv[(32 * TN + 63) / 64];
    int iv[(32 * TM + 63) / 64];
    float fv[(32 * TM + 63) / 64];
#pragma unroll
    for (int u = 0; u < (32 * TN + 63) / 64; ++u) {
      const int c = u * 64 + lane;
      const int64_t cg = cbase + c;
      bv[u] = c >= 32 * TN ? 0.f : (cg < nsp ? (e.bias ? e.bias[cg] : 0.f) : (e.bias2 ? e.bias2[cg - nsp] : 0.f));
    }
#pragma unroll
    for (int u = 0; u < (32 * TM + 63) / 64; ++u) {
      const int r = u * 64 + lane;
      const bool ok = r < 32 * TM;
      iv[u] = !ok ? 0 : (kind == 1 ? (e.rowseg ? e.rowseg[rbase + r] : 0) : (mapped ? (int)e.rng_row[rbase + r] : 0));
      fv[u] = (ok && kind == 1) ? e.rowv[rbase + r] : ((gpair && r < 16 * TN) ? e.gate_wc[(cbase >> 1) + r] : 0.f);    // (gpair: wc of the wave's branch columns)
    }
#pragma unroll
    for (int u = 0; u < (32 * TN + 63) / 64; ++u)
      if (u * 64 + lane < 32 * TN) sbias[u * 64 + lane] = bv[u];
#pragma unroll
    for (int u = 0; u < (32 * TM + 63) / 64; ++u)
      if (u * 64 + lane < 32 * TM) { srow_i[u * 64 + lane] = iv[u]; srow_f[u * 64 + lane] = fv[u]; }
    WAVE_LDS_SYNC();
  }
  float4 ext[4], ext2[4];
  float cs[TN][4];
  float gpa[4][4], gps[TM][4];       // gpair: the pending tanh branch values of a sub-tile; the rows' partial scores
#pragma unroll
  for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
    for (int q = 0; q < 4; ++q) gps[a_][q] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { ext[q] = make_float4(0.f, 0.f, 0.f, 0.f); ext2[q] = make_float4(1.f, 1.f, 1.f, 1.f); }
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int t = 0; t < 4; ++t) cs[b][t] = 0.f;
#define ADVMIL_EPI_PREFETCH(a_, b_)                                                                             \
  do {                                                                                                          \
    if (kind != 0) {                                                                                            \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
        const int rr_ = (a_) * 32 + q * 8 + rq;                                                                 \
        const int64_t xrow_ = kind == 1 ? (int64_t)srow_i[rr_] : rbase + rr_;                                   \
        ext[q] = *reinterpret_cast<const float4*>(xbase + xrow_ * xld + cbase + (b_) * 32 + c4);                \
        if (rmask) ext2[q] = *reinterpret_cast<const float4*>(e.maskref + (rbase + rr_) * (int64_t)e.ldmask + cbase + (b_) * 32 + c4); \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)
  ADVMIL_EPI_PREFETCH(0, 0);
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int64_t col = cbase + b * 32 + c4;
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * PITCH_KC + i] = acc[a][b][r];
      WAVE_LDS_SYNC();
      float res[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v4 = *reinterpret_cast<const float4*>(patch + (q * 8 + rq) * PITCH_KC + c4);
        res[q][0] = v4.x; res[q][1] = v4.y; res[q][2] = v4.z; res[q][3] = v4.w;
      }
      const float4 b4 = *reinterpret_cast<const float4*>(sbias + b * 32 + c4);
      WAVE_LDS_SYNC();   // the reads have landed: the next sub-tile may overwrite the patch
      const int act = gpair ? ((b & 1) ? ACT_SIGMOID : ACT_TANH) : (cbase + b * 32 < e.act_split ? e.act0 : e.act1);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      if (!drop && kind == 0 && !bmask) {
        // plain bias + activation (the forward layers): the launch-uniform tests are taken once per sub-tile, not once per element
        // (16 elements x 5 scalar branches per sub-tile made this path 1.6x slower than its stores alone, tools/probe/store_probe.hip)
        const float al = e.alpha;
        if (act == ACT_RELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) res[q][t] = fmaxf(res[q][t] * al + bb[t], 0.0f);
        } else if (act == ACT_NONE) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) res[q][t] = res[q][t] * al + bb[t];
        } else if (act == ACT_TANH) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) res[q][t] = act_apply(ACT_TANH, res[q][t] * al + bb[t]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) res[q][t] = act_apply(ACT_SIGMOID, res[q][t] * al + bb[t]);
        }
      } else if (!drop && kind == 1 && act == ACT_NONE && !rmask && !bmask) {
        // rank-1 term per bag (dh = dG Wab + A[n] dpooled[bag(n)]), no activation
        const float al = e.alpha;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float r1 = srow_f[a * 32 + q * 8 + rq];
          const float xe[4] = {ext[q].x, ext[q].y, ext[q].z, ext[q].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) res[q][t] = res[q][t] * al + bb[t] + r1 * xe[t];
        }
      } else if (!drop && kind == 1 && act == ACT_NONE && !rmask && bmask) {
        // ... and the bit mask behind it (the first layer's ReLU / dropout backward): same loop, one LDS word per row
        const float al = e.alpha, ms = e.mask_scale;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float r1 = srow_f[a * 32 + q * 8 + rq];
          const uint32_t mw = smask[(a * 32 + q * 8 + rq) * TN + b] >> c4;
          const float xe[4] = {ext[q].x, ext[q].y, ext[q].z, ext[q].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) res[q][t] = ((mw >> t) & 1u) ? (res[q][t] * al + bb[t] + r1 * xe[t]) * ms : 0.0f;
        }
      } else if (drop && kind == 0 && !bmask && (act == ACT_RELU || act == ACT_NONE)) {
        // (ReLU +) dropout from the counter RNG (train-mode forward layers)
        const float al = e.alpha;
        const bool relu = act == ACT_RELU;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t srow = mapped ? (int64_t)srow_i[a * 32 + q * 8 + rq] : rbase + a * 32 + q * 8 + rq;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float x = res[q][t] * al + bb[t];
            x = relu ? fmaxf(x, 0.0f) : x;
            res[q][t] = x * rng_keep(key, (uint64_t)(srow * N + col + t), e.drop_p, inv_keep);
          }
        }
      } else
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xe[4] = {ext[q].x, ext[q].y, ext[q].z, ext[q].w};
        const int64_t srow = mapped ? (int64_t)srow_i[a * 32 + q * 8 + rq] : rbase + a * 32 + q * 8 + rq;
        const float r1 = kind == 1 ? srow_f[a * 32 + q * 8 + rq] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float x = res[q][t] * e.alpha + bb[t];
          if (kind == 1) x += r1 * xe[t];
          x = act_apply(act, x);
          if (drop) x *= rng_keep(key, (uint64_t)(srow * N + col + t), e.drop_p, inv_keep);
          if (kind == 2) x *= (xe[t] > 0.0f ? e.mask_scale : 0.0f);
          if (kind == 3) x += xe[t];
          res[q][t] = x;
        }
        if (rmask) {
          const float me[4] = {ext2[q].x, ext2[q].y, ext2[q].z, ext2[q].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) res[q][t] *= (me[t] > 0.0f ? e.mask_scale : 0.0f);
        }
        if (bmask) {
          const uint32_t mw = smask[(a * 32 + q * 8 + rq) * TN + b] >> c4;
#pragma unroll
          for (int t = 0; t < 4; ++t) res[q][t] *= ((mw >> t) & 1u) ? e.mask_scale : 0.0f;
        }
      }
      if (csum) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < 4; ++t) cs[b][t] += res[q][t];
      }
      if (gpair) {
        // even sub-tile: keep the (dropped) tanh values; odd sub-tile: multiply in the (dropped) sigmoid values and wc, add to the row sums
        const int pu = b >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rloc = a * 32 + q * 8 + rq;
          const uint32_t mw = smask[(rloc * (TN / 2) + pu) * 2 + (b & 1)] >> c4;
          if ((b & 1) == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) gpa[q][t] = ((mw >> t) & 1u) ? res[q][t] * inv_keep : 0.0f;
          } else {
            const float4 w4 = *reinterpret_cast<const float4*>(srow_f + pu * 32 + c4);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) gps[a][q] += ((mw >> t) & 1u) ? gpa[q][t] * (res[q][t] * inv_keep) * wv[t] : 0.0f;
          }
        }
      }
      // next sub-tile's per-element operand: behind this sub-tile's math (the registers are free again), ahead of its stores
      if (b + 1 < TN) ADVMIL_EPI_PREFETCH(a, b + 1);
      else if (a + 1 < TM) ADVMIL_EPI_PREFETCH(a + 1, 0);
      if (PLAIN && cbase + b * 32 >= nsp) {     // second layer's 32 columns (wave-uniform): its own output, no planes
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(e.c2 + (rbase + a * 32 + q * 8 + rq) * e.ldc2 + (col - nsp)) =
              make_float4(res[q][0], res[q][1], res[q][2], res[q][3]);
        continue;
      }
      if (out) {         // (NULL: the caller wants the operand planes of the result only -- advmil_gemm_f32_tiled with C == NULL and c_hi set)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t off = (rbase + a * 32 + q * 8 + rq) * ldo + col;
          *reinterpret_cast<float4*>(out + off) = make_float4(res[q][0], res[q][1], res[q][2], res[q][3]);
        }
      }
      if (e.c_hi) {      // planes of the final values (off % 4 == 0 here: two 8-byte stores per float4)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t off = (rbase + a * 32 + q * 8 + rq) * ldo + col;
          union { __bf16 b[4]; uint2 u; } h, l;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            h.b[t] = (__bf16)res[q][t];
            l.b[t] = (__bf16)(res[q][t] - (float)h.b[t]);
          }
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16raw*>(e.c_hi) + off) = h.u;
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16raw*>(e.c_lo) + off) = l.u;
        }
      }
    }
  }
#undef ADVMIL_EPI_PREFETCH
  if (gpair) {
#pragma unroll
    for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = gps[a_][q];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);     // the 8 lanes that share a row
        if ((lane & 7) == 0) e.gate_out[(rbase + a_ * 32 + q * 8 + rq) * e.gate_np + nt_i * WC + wc] = t;
      }
  }
  if (csum) {
    // the wave's 32 TM rows: the 8 row-lanes (lane >> 3) hold pieces of every column; one partial row per (m tile, wave row)
    const int64_t prow = (m0 / (32 * TM * WR)) * WR + wr;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = cs[b][t];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        cs[b][t] = v;
      }
      if (rq == 0) *reinterpret_cast<float4*>(e.colsum + prow * N + cbase + b * 32 + c4) = make_float4(cs[b][0], cs[b][1], cs[b][2], cs[b][3]);
    }
  }
}

// RAWBAR: the caller has LDS-DMA in flight (persistent plane-fed kernel): the opening barrier must not drain vmcnt.
// Returns a LOWER bound of the vector-memory operations this wave issued and did not wait for (the streaming form's stores), which
// the persistent kernel uses as the count of operations younger than its cross-tile prefetch.
template <int TM, int TN, int WR, int WC, bool RAWBAR = false, int EPI = 0>      // EPI: 0 all forms, 1 no streaming form, 2 streaming form for plain launches only
__device__ __forceinline__ int gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[TM][TN], float* smem, int wave, int lane, int wr, int wc,
                                             int64_t m0, int64_t n0, int z, int nt_i) {
  const int i = lane & 31, hi = lane >> 5;
  const advmil_epilogue_t& e = g.epi;
  const bool direct = (g.splits == 1);
  uint64_t key = 0;
  float inv_keep = 1.0f;
  if (direct && e.seed && e.drop_p > 0.0f) {
    key = rng_key(*e.seed, e.stream_id);
    inv_keep = hw_rcp(1.0f - e.drop_p);
  } else if (direct && e.gate_bits_a && e.drop_p > 0.0f) {
    inv_keep = hw_rcp(1.0f - e.drop_p);      // (gate pair form: the keep bits are given, only the scale is formed here)
  }
  float* const out = direct ? g.C : g.ws + (int64_t)z * g.M * g.N;
  const int64_t ldo = direct ? g.ldc : g.N;
  const bool vec_ok = ((ldo & 3) == 0) && (((uintptr_t)out & 15) == 0);
  float* const patch = smem + wave * EPI_WAVE_FLOATS(TM, TN);
  if constexpr (RAWBAR) {   // every wave is done reading the operand tiles (its fragment reads were consumed by its MFMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();
  }
  // Gate-score mode (e.gate_wc): the columns are the interleaved branches of the gated attention scorer, col 2j = a_j (tanh),
  // col 2j+1 = b_j (sigmoid). Instead of storing C, each row's  sum_j tanh(.)_j * sigmoid(.)_j * wc_j  over this workgroup's
  // columns is reduced in registers / across the 8 lanes of a row and written to gate_out[row * gate_np + column-block]: the
  // no-grad generator pass then never writes (and gate_score never re-reads) the [rows, 2D] activations.
  const bool gate_store = direct && e.gate_wc != nullptr && e.gate_bits_a != nullptr;      // training form: store AND score (streaming epilogue)
  const bool gate_mode = direct && e.gate_wc != nullptr && !gate_store;
  if constexpr (TM * TN >= 4 && EPI != 1) {   // the slab-sized tiles (the 64x64 ... 64x192 tiles serve launch-bound shapes: generic path only)
    const int nmode = (e.rowv && e.maskref && !e.accumulate) ? 1 : (e.rowv ? 1 : 0) + (e.maskref ? 1 : 0) + (e.accumulate ? 1 : 0);
    const bool stream = direct && !gate_mode && vec_ok && (g.N % (32 * TN * WC)) == 0 && (g.M % (32 * TM * WR)) == 0 && (e.act_split & 31) == 0 &&
                        nmode <= 1 && !(e.rowv && e.seed && e.rng_row) && (!e.bias || ((uintptr_t)e.bias & 15) == 0) && (!e.rowv || ((uintptr_t)e.colv & 15) == 0) &&
                        (!e.maskref || ((e.ldmask & 3) == 0 && ((uintptr_t)e.maskref & 15) == 0)) &&
                        (!e.c_hi || ((((uintptr_t)e.c_hi) | ((uintptr_t)e.c_lo)) & 7) == 0) &&
                        (!e.c2 || ((((uintptr_t)e.c2) & 15) == 0 && (e.ldc2 & 3) == 0 && (e.n_split & 31) == 0));
    if (stream) {
      if constexpr (EPI == 2) {
        if (nmode == 0 && !(e.seed && e.drop_p > 0.0f)) {
          gemm_epilogue_stream<TM, TN, WR, WC, true>(g, acc, patch, lane, wr, wc, m0, n0, key, inv_keep, nt_i);
          return TM * TN * 4;
        }
      } else {
        gemm_epilogue_stream<TM, TN, WR, WC>(g, acc, patch, lane, wr, wc, m0, n0, key, inv_keep);
        return nmode == 0 ? TM * TN * 4 : 0;      // (with a prefetched per-element operand the stores are partly waited for)
      }
    }
  }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < TN; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * PITCH_KC + i] = acc[a][b][r];
      WAVE_LDS_SYNC();   // the patch is private to this wave: order its LDS writes before the reads below, no block barrier
      const int64_t rbase = m0 + wr * 32 * TM + a * 32;
      if (gate_mode) {
        const int64_t col = n0 + wc * 32 * TN + b * 32 + (lane & 7) * 4;      // N % 4 == 0 in this mode: whole float4 or nothing
        float w0 = 0.f, w1 = 0.f;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < g.N) {
          w0 = e.gate_wc[col >> 1]; w1 = e.gate_wc[(col >> 1) + 1];
          if (e.bias) b4 = *reinterpret_cast<const float4*>(e.bias + col);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pr = q * 8 + (lane >> 3);
          const float4 v4 = *reinterpret_cast<const float4*>(patch + pr * PITCH_KC + (lane & 7) * 4);
          const float t0 = act_apply(ACT_TANH, v4.x * e.alpha + b4.x), s0 = act_apply(ACT_SIGMOID, v4.y * e.alpha + b4.y);
          const float t1 = act_apply(ACT_TANH, v4.z * e.alpha + b4.z), s1 = act_apply(ACT_SIGMOID, v4.w * e.alpha + b4.w);
          gsum[q] += t0 * s0 * w0 + t1 * s1 * w1;
        }
        WAVE_LDS_SYNC();
        continue;
      }
      // a lane stores columns col..col+3 of rows rbase + (lane >> 3) + 8q: everything that depends on the column only (bias,
      // which activation) is fetched once per sub-tile, not once per element inside the row loop
      const int64_t col = n0 + wc * 32 * TN + b * 32 + (lane & 7) * 4;
      const int nvalid = col >= g.N ? 0 : ((g.N - col >= 4) ? 4 : (int)(g.N - col));
      float bias4[4] = {0.f, 0.f, 0.f, 0.f};
      int act4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (direct && e.bias && t < nvalid) bias4[t] = e.bias[col + t];
        act4[t] = (col + t) < e.act_split ? e.act0 : e.act1;
      }
      const bool same_act = act4[0] == act4[3];
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const int pr = q * 8 + (lane >> 3);
        const int64_t row = rbase + pr;
        const float4 v4 = *reinterpret_cast<const float4*>(patch + pr * PITCH_KC + (lane & 7) * 4);
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        if (row < g.M && nvalid > 0) {
          float* c = out + row * ldo + col;
          if (direct) {
            float r1 = 0.f;
            const float* cv = nullptr;
            const int64_t grow = (e.seed && e.rng_row) ? e.rng_row[row] : row;      // the row's index in the dropout stream
            if (e.rowv) {
              r1 = e.rowv[row];
              cv = e.colv + (e.rowseg ? (int64_t)e.rowseg[row] * g.N : 0) + col;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (t < nvalid) {
                float x = v[t] * e.alpha + bias4[t];
                if (cv) x += r1 * cv[t];
                x = act_apply(same_act ? act4[0] : act4[t], x);
                if (e.seed && e.drop_p > 0.0f) x *= rng_keep(key, (uint64_t)(grow * g.N + col + t), e.drop_p, inv_keep);
                if (e.maskref) x *= (e.maskref[row * (int64_t)e.ldmask + col + t] > 0.0f ? e.mask_scale : 0.0f);
                if (e.accumulate) x += c[t];
                v[t] = x;
              }
          }
          if (!out) {
            // planes only (direct launches with c_hi: checked by the host side)
          } else if (nvalid == 4 && vec_ok) {
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)      // static indices: a runtime-bounded loop would push v[] into scratch memory
              if (t < nvalid) c[t] = v[t];
          }
          if (direct && e.c_hi) emit_planes4(e, row * ldo + col, v, nvalid);
        }
      }
      WAVE_LDS_SYNC();   // reads of this patch done before the next sub-tile overwrites it
    }
    if (gate_mode) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = gsum[q];
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);     // the 8 lanes that share a row
        const int64_t row = m0 + wr * 32 * TM + a * 32 + q * 8 + (lane >> 3);
        if ((lane & 7) == 0 && row < g.M) e.gate_out[row * e.gate_np + nt_i * WC + wc] = t;
      }
    }
  }
  return 0;
}

// ---- shared by the two plane-fed kernels: the LDS-DMA's global address space and counted vector-memory waits
#define GLB_AS __attribute__((address_space(1)))
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
