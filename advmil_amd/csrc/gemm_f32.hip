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
                                                     int64_t m0, int64_t n0, uint64_t key, float inv_keep) {
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
  const float* const xbase = kind == 1 ? e.colv : (kind == 2 ? e.maskref : out);
  const int64_t xld = kind == 1 ? N : (kind == 2 ? (int64_t)e.ldmask : ldo);
  const int64_t rbase = m0 + wr * 32 * TM, cbase = n0 + wc * 32 * TN;
  const int64_t nsp = (PLAIN && e.c2) ? e.n_split : N;          // columns >= nsp belong to the launch's second layer (two-layer form)
  // ---- side data -> LDS: sbias[32*TN] | srow_i[32*TM] (bag index, or dropout stream row: never both in one launch) | srow_f[32*TM]
  float* const sbias = patch + 32 * PITCH_KC;
  int* const srow_i = reinterpret_cast<int*>(sbias + 32 * TN);
  float* const srow_f = sbias + 32 * TN + 32 * TM;
  uint32_t* const smask = reinterpret_cast<uint32_t*>(sbias + 32 * TN + 64 * TM);      // [32 TM rows][TN words]
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
      fv[u] = (ok && kind == 1) ? e.rowv[rbase + r] : 0.f;
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
      const int act = cbase + b * 32 < e.act_split ? e.act0 : e.act1;
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
  const bool gate_mode = direct && e.gate_wc != nullptr;
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
          gemm_epilogue_stream<TM, TN, WR, WC, true>(g, acc, patch, lane, wr, wc, m0, n0, key, inv_keep);
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
#define GLB_AS __attribute__((address_space(1)))
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

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
    if (!epi->a_lo) {                                     // A = a bf16 slab (x_storage = "bf16"): two products per MFMA step
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
    if (!epi->b_lo) {
      switch (tile) {
        case 91: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 2, 4, 3, 0>), tgrid, dim3(512), 0, stream, g); break;
        case 92: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 4, 2, 3, 0>), tgrid, dim3(512), 0, stream, g); break;
        default: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 4, 4, 2, 2, 0>), tgrid, dim3(512), 0, stream, g); break;
      }
    } else
    switch (tile) {
      case 91: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 2, 4, 3>), tgrid, dim3(512), 0, stream, g); break;   // 3 x 48 KB
      case 92: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 2, 4, 2, 3>), tgrid, dim3(512), 0, stream, g); break;   // 3 x 48 KB
      default: hipLaunchKernelGGL((gemm_tn_planes_kernel<2, 4, 4, 2, 2>), tgrid, dim3(512), 0, stream, g); break;   // 2 x 64 KB
    }
    ADVMIL_LAUNCH_CHECK();
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
