// Fused self-attention core of the ESAT layer for gfx950 (nn.MultiheadAttention inside nn.TransformerEncoderLayer,
// /root/reference model/backbone_utils.py:113-127, called from DualTrans_HS.forward, model/backbone.py:188-196):
//   O = dropout(softmax(Q K^T / sqrt(hd))) V   per (bag, head), non-causal, bags never attend across their boundary.
// Flash-style: the [L, L] score matrix never exists in HBM. Forward keeps a running (max, sum) per query and one log-sum-exp per
// (query, head) for the backward; the backward recomputes the probabilities tile by tile in two launches (dQ with the queries
// stationary, dK/dV with the keys stationary) so that no gradient needs a float atomic: results are deterministic.
//
// Arithmetic: "bf16x3" -- every fp32 operand element is split hi + lo (bf16 each) and a product is three
// v_mfma_f32_32x32x16_bf16 (lo.hi + hi.lo + hi.hi) with fp32 accumulation, the same arithmetic as the contraction engine's
// bf16x3 mode (gemm_f32.hip); softmax statistics, exponentials and the rescaling are fp32.
//
// Layout per workgroup (4 waves, 256 threads): wave w owns 32 rows of the stationary operand (queries for fwd / dQ, keys for dK/dV)
// as MFMA *columns* (B operand, fragments resident in VGPRs); the streamed operand comes through LDS in tiles of 64 rows as two
// bf16 planes (hi, lo) of [row][72 halfwords]:
//   * 144-byte pitch: a ds_read_b128 row fragment (lane = row, 8 consecutive head dims) puts the 16 lanes of a group on 16
//     distinct 16-byte bank slots, and the four rows of a ds_read_b64_tr_b16 block land on four distinct 32-byte bank windows;
//   * contractions over the head dimension read row fragments (ds_read_b128);
//   * contractions over the streamed rows (P.V, dS^T.K, P^T.dO, dS^T.Q) read the SAME planes through the LDS transpose read,
//     and take their second operand straight from the accumulator registers of the score tile: with scores computed transposed
//     (rows = streamed rows, column = lane's stationary row) a lane holds, for its column, 8 row values per 16-row k-step in
//     exactly the (lane-half, slot) positions an MFMA B fragment wants once the k-slot <-> row map is chosen as
//     slot t of half h  <->  row 16*s + 8*(t>>2) + 4*h + (t&3); the transpose reads use the same map, so no cross-lane traffic.
//   * per-query softmax statistics are lane-local (column = query); the two lane halves exchange one max per tile.
// head_dim 48 = 3 k-steps of 16 for Q K^T (no padding); the 48 output dims of P.V / dQ / dK / dV occupy 1.5 MFMA row tiles
// (pad columns of the planes are zero).
// Workgroup -> (bag, tile, head) with head = blockIdx % nhead: for nhead = 8 every XCD (blockIdx % 8) serves one head, so the K/V
// panel of a (bag, head) is fetched into exactly one L2.
//
// Dropout on the attention probabilities (train mode): keep(i, j) = 16-bit half (j & 1) of hash32(rowkey(i) + (j >> 1) * 0x9E3779B9)
// >= floor(p * 2^16), rowkey(i) = high word of splitmix64(key(seed, stream) + (global region row of query i) * nhead + head);
// restated on the host in advmil_amd/synth.py::attn_dropout_keep. A per-row 64-bit mix + one 32-bit finaliser per key PAIR costs
// ~5 VALU per probability where a lane walks keys (forward, dQ) -- a splitmix64 per element would cost ~4x the MFMA time of the
// tile -- and is layout independent, which the key-stationary backward needs (there a lane walks queries and hashes per element).
// The drop probability is thereby quantised to 1/65536 (0.25 is exact).
#include "common.h"
#include "bf16split.h"
#include "../../include/advmil_hip.h"

#define AT_PITCH 72   // halfwords per plane row
#define AT_KT 64      // streamed rows per LDS tile
// stationary rows per workgroup = 32 * NW (NW = 4 or 8 waves): a (bag, head)'s streamed K/V (or Q/dO) panel is re-read by every
// such workgroup, so 8 waves halve that traffic and the staging work per stationary row; the host takes 8 when the bags are long
#define AT_PLANE (AT_KT * AT_PITCH)

struct AttnArgs {
  const float* qkv;   // [Ltot, 3*H*HD] packed in-projection output (q | k | v), row pitch ldq
  int64_t ldq;
  float* out;         // fwd: O [Ltot, H*HD]
  float* lse;         // [Ltot, H] log2-domain log-sum-exp of the scaled scores
  const float* dout;  // bwd: dO [Ltot, H*HD]
  const float* dsum;  // bwd: D [Ltot, H] = sum_d dO * O
  float* dqkv;        // bwd: [Ltot, 3*H*HD], row pitch ldq
  const int64_t* ptr; // [nseg+1] first region row of every bag, or NULL (one bag of Ltot rows)
  const int64_t* rng_rowoff;  // [nseg] added to a bag's local rows to form the dropout stream's row id, or NULL
  int64_t Ltot;
  int nseg, ntile, H;
  float scale_log2e;  // log2(e) / sqrt(head_dim)
  float scale;        // 1 / sqrt(head_dim)
  uint32_t drop_thr;  // floor(p * 2^16)
  float inv_keep;     // 1 / (1 - p)
  const uint64_t* seed;
  uint64_t stream_id;
};

__device__ __forceinline__ uint32_t attn_row_key(uint64_t key, uint64_t row_id) { return (uint32_t)(splitmix64(key + row_id) >> 32); }
// one 32-bit hash serves the key PAIR (2m, 2m+1): 16 bits each, compared with floor(p * 2^16)
__device__ __forceinline__ uint32_t attn_hash(uint32_t rk, uint32_t jpair) {
  uint32_t x = rk + jpair * 0x9E3779B9u;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool attn_keep(uint32_t rk, uint32_t j, uint32_t thr) {
  return ((attn_hash(rk, j >> 1) >> ((j & 1u) * 16u)) & 0xffffu) >= thr;
}

// ---- staging: [64 rows][HD] fp32 from global -> registers -> two bf16 planes in LDS
template <int HD, int NT>
struct TileRegs {
  static constexpr int NE = AT_KT * (HD / 4);          // float4 elements of a streamed tile
  static constexpr int NP = (NE + NT - 1) / NT;
  float4 f[NP];
  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t row_end, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int e = p * NT + tid;
      const int64_t row = row0 + e / (HD / 4);
      const int c4 = e % (HD / 4);
      f[p] = (e < NE && row < row_end) ? *reinterpret_cast<const float4*>(base + row * ld + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(bf16raw* __restrict__ planes, int tid, float mul) const {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int e = p * NT + tid;
      if (NE % NT != 0 && e >= NE) break;
      const int off = (e / (HD / 4)) * AT_PITCH + (e % (HD / 4)) * 4;
      uint2 h, l;
      split4(make_float4(f[p].x * mul, f[p].y * mul, f[p].z * mul, f[p].w * mul), h, l);
      *reinterpret_cast<uint2*>(planes + off) = h;
      *reinterpret_cast<uint2*>(planes + AT_PLANE + off) = l;
    }
  }
};

// row fragment: lane (i = lane & 31, half = lane >> 5) <- plane[row][koff .. koff + 7]
__device__ __forceinline__ bf16x8 frag_rows(const bf16raw* __restrict__ plane, int row, int koff) {
  Frag8 f;
  f.u = *reinterpret_cast<const uint4*>(plane + row * AT_PITCH + koff);
  return f.v;
}
// transposed fragment: lane (i = lane & 31 -> column mbase + i, half) <- rows r0 + {0..3} (slots 0-3) and r0 + 8 + {0..3} (slots 4-7);
// r0 already holds the half's offset (16*s + 4*half): the k-slot <-> row map of the header comment
__device__ __forceinline__ bf16x8 frag_tr(const bf16raw* __restrict__ plane, int r0, int mbase, int lane) {
  const bf16raw* p = plane + (r0 + ((lane & 15) >> 2)) * AT_PITCH + mbase + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
  union { bf16x4_t q[2]; bf16x8 v; } a;
  a.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p));
  a.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)(p + 8 * AT_PITCH));
  return a.v;
}
// 8 fp32 accumulator values -> hi / lo B fragments
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  union { unsigned u[4]; bf16x8 v; } h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) split2(v[2 * i], v[2 * i + 1], h.u[i], l.u[i]);
  hi = h.v; lo = l.v;
}
// the stationary operand's fragments: 8 consecutive head dims of one row at 16*ks + 8*half, scaled, split
template <int HD>
__device__ __forceinline__ void load_row_frags(const float* __restrict__ row, bool ok, int half, float mul, bf16x8 (&fh)[HD / 16],
                                               bf16x8 (&fl)[HD / 16]) {
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (ok) {
      a = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * half);
      b = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * half + 4);
    }
    const float v[8] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul, b.x * mul, b.y * mul, b.z * mul, b.w * mul};
    split8(v, fh[ks], fl[ks]);
  }
}
#define MFMA3(acc, ah, al, bh, bl)                                        \
  do {                                                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);  \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);  \
  } while (0)

__device__ __forceinline__ void zero_lds(bf16raw* smem, int halfwords, int tid, int nt) {
  for (int e = tid * 8; e < halfwords; e += nt * 8) *reinterpret_cast<uint4*>(smem + e) = make_uint4(0u, 0u, 0u, 0u);
}
// row of accumulator register r in a 32x32 tile: (r & 3) + 8 * (r >> 2) + 4 * half
#define ACC_ROW(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// =====================================================================================
// forward
// =====================================================================================
template <int HD, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(AttnArgs a) {
  constexpr int NT = 64 * NW, AT_QB = 32 * NW;
  constexpr int KS = HD / 16, DT = (HD + 31) / 32;
  __shared__ __attribute__((aligned(16))) bf16raw smem[4 * AT_PLANE];   // K hi | K lo | V hi | V lo
  bf16raw* const sK = smem;
  bf16raw* const sV = smem + 2 * AT_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, half = lane >> 5;
  const int H = a.H;
  const int h = blockIdx.x % H;
  const int rest = blockIdx.x / H;
  const int qt = rest % a.ntile, g = rest / a.ntile;
  const int64_t row0 = a.ptr ? a.ptr[g] : 0;
  const int64_t Lg = a.ptr ? a.ptr[g + 1] - row0 : a.Ltot;
  if ((int64_t)qt * AT_QB >= Lg) return;
  const int D = H * HD;
  const float* const Qb = a.qkv + row0 * a.ldq + h * HD;
  const float* const Kb = Qb + D;
  const float* const Vb = Qb + 2 * D;
  zero_lds(smem, 4 * AT_PLANE, tid, NT);   // pad columns [HD, 72) stay zero: the transposed reads of the last head-dim tile cover [32, 64)

  const int64_t q = (int64_t)qt * AT_QB + wave * 32 + j;
  const bool qok = q < Lg;
  bf16x8 qh[KS], ql[KS];
  load_row_frags<HD>(Qb + q * a.ldq, qok, half, a.scale_log2e, qh, ql);
  uint32_t rk = 0;
  if (DROP) {
    const uint64_t grow = (uint64_t)(row0 + (a.rng_rowoff ? a.rng_rowoff[g] : 0) + q);
    rk = attn_row_key(rng_key(*a.seed, a.stream_id), grow * (uint64_t)H + (uint64_t)h);
  }

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  TileRegs<HD, NT> rK, rV;
  rK.load(Kb, a.ldq, 0, Lg, tid);
  rV.load(Vb, a.ldq, 0, Lg, tid);
  for (int64_t kb = 0; kb < Lg; kb += AT_KT) {
    __syncthreads();                       // every wave is done with the previous tile (and the zero fill)
    rK.store(sK, tid, 1.f);
    rV.store(sV, tid, 1.f);
    __syncthreads();
    if (kb + AT_KT < Lg) {                 // next tile's loads fly under this tile's MFMAs
      rK.load(Kb, a.ldq, kb + AT_KT, Lg, tid);
      rV.load(Vb, a.ldq, kb + AT_KT, Lg, tid);
    }
    // ---- S^T[key, q] = K . Q'^T  (already in the log2 domain)
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kh = frag_rows(sK, 32 * t + j, 16 * ks + 8 * half);
        const bf16x8 kl = frag_rows(sK + AT_PLANE, 32 * t + j, 16 * ks + 8 * half);
        MFMA3(s[t], kh, kl, qh[ks], ql[ks]);
      }
    }
    if (kb + AT_KT > Lg) {                 // ragged tail: keys past the bag get probability 0
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb + 32 * t + ACC_ROW(r, half) >= Lg) s[t][r] = -INFINITY;
    }
    float mx = m_run;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // rescale the running state only when some query of this wave saw a new maximum (after the first tiles it rarely moves)
    const bool moved = __any(mx != m_run);
    float alpha = 1.f;
    if (moved) {
      alpha = hw_exp2(m_run - mx);
      m_run = mx;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = hw_exp2(s[t][r] - mx);
        psum += p;
        s[t][r] = p;
      }
    if (DROP) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {       // 4 consecutive keys per register group = 2 hashes
          const uint32_t jp = (uint32_t)(kb + 32 * t + 8 * rg + 4 * half) >> 1;
          const uint32_t h0 = attn_hash(rk, jp), h1 = attn_hash(rk, jp + 1);
          if ((h0 & 0xffffu) < a.drop_thr) s[t][4 * rg] = 0.f;
          if ((h0 >> 16) < a.drop_thr) s[t][4 * rg + 1] = 0.f;
          if ((h1 & 0xffffu) < a.drop_thr) s[t][4 * rg + 2] = 0.f;
          if ((h1 >> 16) < a.drop_thr) s[t][4 * rg + 3] = 0.f;
        }
    }
    l_run = l_run * alpha + psum;
    // ---- O^T[d, q] += V^T[d, key] . P^T[key, q]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s[t][8 * s2 + u];
        bf16x8 ph, pl;
        split8(v, ph, pl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 vh = frag_tr(sV, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          const bf16x8 vl = frag_tr(sV + AT_PLANE, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          MFMA3(o[dt], vh, vl, ph, pl);
        }
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qok) {
    const float inv = (DROP ? a.inv_keep : 1.f) * hw_rcp(l_tot);
    float* const orow = a.out + (row0 + q) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * dt + 8 * rg + 4 * half;
        if (d < HD)
          *reinterpret_cast<float4*>(orow + d) =
              make_float4(o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
      }
    if (half == 0) a.lse[(row0 + q) * H + h] = m_run + hw_log2(l_tot);
  }
}

// =====================================================================================
// backward, queries stationary: dQ = scale * dS K
// =====================================================================================
template <int HD, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int NT = 64 * NW, AT_QB = 32 * NW;
  constexpr int KS = HD / 16, DT = (HD + 31) / 32;
  __shared__ __attribute__((aligned(16))) bf16raw smem[4 * AT_PLANE];   // K hi | K lo | V hi | V lo
  bf16raw* const sK = smem;
  bf16raw* const sV = smem + 2 * AT_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, half = lane >> 5;
  const int H = a.H;
  const int h = blockIdx.x % H;
  const int rest = blockIdx.x / H;
  const int qt = rest % a.ntile, g = rest / a.ntile;
  const int64_t row0 = a.ptr ? a.ptr[g] : 0;
  const int64_t Lg = a.ptr ? a.ptr[g + 1] - row0 : a.Ltot;
  if ((int64_t)qt * AT_QB >= Lg) return;
  const int D = H * HD;
  const float* const Qb = a.qkv + row0 * a.ldq + h * HD;
  const float* const Kb = Qb + D;
  const float* const Vb = Qb + 2 * D;
  zero_lds(smem, 4 * AT_PLANE, tid, NT);

  const int64_t q = (int64_t)qt * AT_QB + wave * 32 + j;
  const bool qok = q < Lg;
  bf16x8 qh[KS], ql[KS], gh[KS], gl[KS];
  load_row_frags<HD>(Qb + q * a.ldq, qok, half, a.scale_log2e, qh, ql);
  load_row_frags<HD>(a.dout + (row0 + q) * D + h * HD, qok, half, 1.f, gh, gl);
  const float lse_q = qok ? a.lse[(row0 + q) * H + h] : 0.f;
  const float d_q = qok ? a.dsum[(row0 + q) * H + h] : 0.f;
  uint32_t rk = 0;
  if (DROP) {
    const uint64_t grow = (uint64_t)(row0 + (a.rng_rowoff ? a.rng_rowoff[g] : 0) + q);
    rk = attn_row_key(rng_key(*a.seed, a.stream_id), grow * (uint64_t)H + (uint64_t)h);
  }
  const float ik = DROP ? a.inv_keep : 1.f;

  f32x16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  TileRegs<HD, NT> rK, rV;
  rK.load(Kb, a.ldq, 0, Lg, tid);
  rV.load(Vb, a.ldq, 0, Lg, tid);
  for (int64_t kb = 0; kb < Lg; kb += AT_KT) {
    __syncthreads();
    rK.store(sK, tid, 1.f);
    rV.store(sV, tid, 1.f);
    __syncthreads();
    if (kb + AT_KT < Lg) {
      rK.load(Kb, a.ldq, kb + AT_KT, Lg, tid);
      rV.load(Vb, a.ldq, kb + AT_KT, Lg, tid);
    }
    // one 32-key sub-tile at a time through scores -> dS -> dQ: only one pair of 32x32 accumulators is live (both sub-tiles at once
    // put the dropout instantiation 5 registers over the file: 20 bytes of scratch)
    const bool tail = kb + AT_KT > Lg;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kh = frag_rows(sK, 32 * t + j, 16 * ks + 8 * half);
        const bf16x8 kl = frag_rows(sK + AT_PLANE, 32 * t + j, 16 * ks + 8 * half);
        MFMA3(st, kh, kl, qh[ks], ql[ks]);                   // S^T[key, q]
        const bf16x8 vh = frag_rows(sV, 32 * t + j, 16 * ks + 8 * half);
        const bf16x8 vl = frag_rows(sV + AT_PLANE, 32 * t + j, 16 * ks + 8 * half);
        MFMA3(dpt, vh, vl, gh[ks], gl[ks]);                  // dPd^T[key, q] = V . dO^T
      }
      if (DROP) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const uint32_t jp = (uint32_t)(kb + 32 * t + 8 * rg + 4 * half) >> 1;
          const uint32_t h0 = attn_hash(rk, jp), h1 = attn_hash(rk, jp + 1);
          if ((h0 & 0xffffu) < a.drop_thr) dpt[4 * rg] = 0.f;
          if ((h0 >> 16) < a.drop_thr) dpt[4 * rg + 1] = 0.f;
          if ((h1 & 0xffffu) < a.drop_thr) dpt[4 * rg + 2] = 0.f;
          if ((h1 >> 16) < a.drop_thr) dpt[4 * rg + 3] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t key = kb + 32 * t + ACC_ROW(r, half);
        float p = hw_exp2(st[r] - lse_q);
        if (tail && key >= Lg) p = 0.f;
        st[r] = p * (dpt[r] * ik - d_q);                     // dS^T
      }
      // dQ^T[d, q] += K^T[d, key] . dS^T[key, q]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = st[8 * s2 + u];
        bf16x8 dh, dl;
        split8(v, dh, dl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 kh = frag_tr(sK, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          const bf16x8 kl = frag_tr(sK + AT_PLANE, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          MFMA3(dq[dt], kh, kl, dh, dl);
        }
      }
    }
  }
  if (qok) {
    float* const drow = a.dqkv + (row0 + q) * a.ldq + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * dt + 8 * rg + 4 * half;
        if (d < HD)
          *reinterpret_cast<float4*>(drow + d) = make_float4(dq[dt][4 * rg] * a.scale, dq[dt][4 * rg + 1] * a.scale,
                                                             dq[dt][4 * rg + 2] * a.scale, dq[dt][4 * rg + 3] * a.scale);
      }
  }
}

// =====================================================================================
// backward, keys stationary: dV = Pd^T dO, dK = scale * dS^T Q
// =====================================================================================
template <int HD, bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_dkv_kernel(AttnArgs a) {
  constexpr int NT = 64 * NW, AT_QB = 32 * NW;
  constexpr int KS = HD / 16, DT = (HD + 31) / 32;
  // Q' hi | Q' lo | dO hi | dO lo | lse[64] | D[64] | rowkey[64]
  __shared__ __attribute__((aligned(16))) bf16raw smem[4 * AT_PLANE + 3 * AT_KT * 2];
  bf16raw* const sQ = smem;
  bf16raw* const sG = smem + 2 * AT_PLANE;
  float* const sLse = reinterpret_cast<float*>(smem + 4 * AT_PLANE);
  float* const sD = sLse + AT_KT;
  uint32_t* const sRk = reinterpret_cast<uint32_t*>(sD + AT_KT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, half = lane >> 5;
  const int H = a.H;
  const int h = blockIdx.x % H;
  const int rest = blockIdx.x / H;
  const int kt = rest % a.ntile, g = rest / a.ntile;
  const int64_t row0 = a.ptr ? a.ptr[g] : 0;
  const int64_t Lg = a.ptr ? a.ptr[g + 1] - row0 : a.Ltot;
  if ((int64_t)kt * AT_QB >= Lg) return;
  const int D = H * HD;
  const float* const Qb = a.qkv + row0 * a.ldq + h * HD;
  const float* const Kb = Qb + D;
  const float* const Vb = Qb + 2 * D;
  const float* const Gb = a.dout + row0 * D + h * HD;
  zero_lds(smem, 4 * AT_PLANE, tid, NT);

  const int64_t key = (int64_t)kt * AT_QB + wave * 32 + j;
  const bool kok = key < Lg;
  bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
  load_row_frags<HD>(Kb + key * a.ldq, kok, half, 1.f, kh, kl);
  load_row_frags<HD>(Vb + key * a.ldq, kok, half, 1.f, vh, vl);
  const uint64_t key64 = DROP ? rng_key(*a.seed, a.stream_id) : 0;
  const int64_t rowoff = a.rng_rowoff ? a.rng_rowoff[g] : 0;
  const float ik = DROP ? a.inv_keep : 1.f;

  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  TileRegs<HD, NT> rQ, rG;
  rQ.load(Qb, a.ldq, 0, Lg, tid);
  rG.load(Gb, D, 0, Lg, tid);
  for (int64_t qb = 0; qb < Lg; qb += AT_KT) {
    __syncthreads();
    rQ.store(sQ, tid, a.scale_log2e);
    rG.store(sG, tid, 1.f);
    if (tid < AT_KT) {
      const int64_t qq = qb + tid;
      const bool ok = qq < Lg;
      sLse[tid] = ok ? a.lse[(row0 + qq) * H + h] : 0.f;
      sD[tid] = ok ? a.dsum[(row0 + qq) * H + h] : 0.f;
      if (DROP) sRk[tid] = attn_row_key(key64, (uint64_t)(row0 + rowoff + qq) * (uint64_t)H + (uint64_t)h);
    }
    __syncthreads();
    if (qb + AT_KT < Lg) {
      rQ.load(Qb, a.ldq, qb + AT_KT, Lg, tid);
      rG.load(Gb, D, qb + AT_KT, Lg, tid);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 ah = frag_rows(sQ, 32 * t + j, 16 * ks + 8 * half);
        const bf16x8 al = frag_rows(sQ + AT_PLANE, 32 * t + j, 16 * ks + 8 * half);
        MFMA3(s, ah, al, kh[ks], kl[ks]);                    // S[q, key]
        const bf16x8 bh = frag_rows(sG, 32 * t + j, 16 * ks + 8 * half);
        const bf16x8 bl = frag_rows(sG + AT_PLANE, 32 * t + j, 16 * ks + 8 * half);
        MFMA3(dp, bh, bl, vh[ks], vl[ks]);                   // dPd[q, key] = dO . V^T
      }
      // rows of this tile are queries: 4 consecutive ones per register group
      float pd[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int qo = 32 * t + 8 * rg + 4 * half;
        const float4 l4 = *reinterpret_cast<const float4*>(sLse + qo);
        const float4 d4 = *reinterpret_cast<const float4*>(sD + qo);
        uint4 k4 = make_uint4(0u, 0u, 0u, 0u);
        if (DROP) k4 = *reinterpret_cast<const uint4*>(sRk + qo);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
        const uint32_t kv[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = 4 * rg + u;
          const float p = hw_exp2(s[r] - lv[u]);
          const bool keep = !DROP || attn_keep(kv[u], (uint32_t)key, a.drop_thr);
          pd[r] = keep ? p * ik : 0.f;
          s[r] = p * ((keep ? dp[r] * ik : 0.f) - dvv[u]);    // dS[q, key]
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { v[u] = pd[8 * s2 + u]; w[u] = s[8 * s2 + u]; }
        bf16x8 ph, pl, dh, dl;
        split8(v, ph, pl);
        split8(w, dh, dl);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 gh = frag_tr(sG, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          const bf16x8 gl = frag_tr(sG + AT_PLANE, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          MFMA3(dv[dt], gh, gl, ph, pl);                     // dV^T[d, key] += dO^T[d, q] . Pd[q, key]
          const bf16x8 qh = frag_tr(sQ, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          const bf16x8 ql = frag_tr(sQ + AT_PLANE, 32 * t + 16 * s2 + 4 * half, 32 * dt, lane);
          MFMA3(dk[dt], qh, ql, dh, dl);                     // dK^T[d, key] += Q'^T[d, q] . dS[q, key]
        }
      }
    }
  }
  if (kok) {
    // Q' carries scale * log2(e): dK = scale * dS^T Q = (dS^T Q') * ln 2
    const float ln2 = 0.693147180559945309f;
    float* const krow = a.dqkv + (row0 + key) * a.ldq + D + h * HD;
    float* const vrow = krow + D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 32 * dt + 8 * rg + 4 * half;
        if (d < HD) {
          *reinterpret_cast<float4*>(krow + d) =
              make_float4(dk[dt][4 * rg] * ln2, dk[dt][4 * rg + 1] * ln2, dk[dt][4 * rg + 2] * ln2, dk[dt][4 * rg + 3] * ln2);
          *reinterpret_cast<float4*>(vrow + d) = make_float4(dv[dt][4 * rg], dv[dt][4 * rg + 1], dv[dt][4 * rg + 2], dv[dt][4 * rg + 3]);
        }
      }
  }
}

// D[row, h] = sum_d dO[row, h*HD + d] * O[row, h*HD + d]   (the softmax backward's row constant; holds with dropout, since
// sum_j dP_j P_j = sum_j dPd_j Pd_j = dO . O)
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            int64_t n /* rows * H */, float* __restrict__ dsum) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float4* a = reinterpret_cast<const float4*>(dout + idx * HD);
  const float4* b = reinterpret_cast<const float4*>(out + idx * HD);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const float4 x = a[c], y = b[c];
    s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
  }
  dsum[idx] = s;
}

// =====================================================================================
// C ABI
// =====================================================================================
static int attn_args(AttnArgs& a, const float* qkv, int64_t Ltot, int nhead, int head_dim, int nseg, const int64_t* ptr,
                     int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_rowoff, int& nw) {
  if (!qkv || Ltot <= 0 || nhead <= 0 || nseg <= 0 || max_len <= 0 || max_len > Ltot) return ADVMIL_EINVAL;
  if (head_dim != 48) return ADVMIL_EINVAL;                 // the ESAT layer: d_model 384 / 8 heads (model/backbone.py:30-33)
  if (nseg > 1 && !ptr) return ADVMIL_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f) return ADVMIL_EINVAL;
  if ((uintptr_t)qkv & 15) return ADVMIL_EINVAL;
  // 8-wave workgroups (256 stationary rows) when the bags are long enough to still fill the chip: half the streamed-panel re-reads
  static const int force_nw = []() { const char* e = getenv("ADVMIL_ATTN_WAVES"); return e ? atoi(e) : 0; }();
  nw = (force_nw == 4 || force_nw == 8) ? force_nw : ((max_len >= 1024 && (max_len / 256) * nseg * nhead >= 512) ? 8 : 4);
  const int64_t ntile = (max_len + 32 * nw - 1) / (32 * nw);
  if (ntile * nseg * nhead > 0x7fffffffLL) return ADVMIL_EINVAL;
  a.qkv = qkv; a.ldq = 3 * (int64_t)nhead * head_dim;
  a.out = nullptr; a.lse = nullptr; a.dout = nullptr; a.dsum = nullptr; a.dqkv = nullptr;
  a.ptr = ptr; a.rng_rowoff = rng_rowoff; a.Ltot = Ltot; a.nseg = nseg; a.ntile = (int)ntile; a.H = nhead;
  a.scale = 1.0f / sqrtf((float)head_dim);
  a.scale_log2e = a.scale * 1.44269504088896340736f;
  const bool drop = seed && drop_p > 0.f;
  a.seed = drop ? seed : nullptr;
  a.stream_id = stream_id;
  a.drop_thr = drop ? (uint32_t)((double)drop_p * 65536.0) : 0u;
  a.inv_keep = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  return ADVMIL_OK;
}

extern "C" int advmil_mha_fwd(const float* qkv, int64_t Ltot, int nhead, int head_dim, int nseg, const int64_t* ptr,
                              int64_t max_len, float drop_p, const uint64_t* seed, uint64_t stream_id,
                              const int64_t* rng_rowoff, float* out, float* lse, advmil_stream_t stream_) {
  AttnArgs a;
  int nw = 4;
  const int rc = attn_args(a, qkv, Ltot, nhead, head_dim, nseg, ptr, max_len, drop_p, seed, stream_id, rng_rowoff, nw);
  if (rc) return rc;
  if (!out || !lse || ((uintptr_t)out & 15)) return ADVMIL_EINVAL;
  a.out = out; a.lse = lse;
  const dim3 grid((unsigned)(a.ntile * nseg * nhead));
  if (nw == 8) {
    if (a.seed) hipLaunchKernelGGL((attn_fwd_kernel<48, true, 8>), grid, dim3(512), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<48, false, 8>), grid, dim3(512), 0, (hipStream_t)stream_, a);
  } else {
    if (a.seed) hipLaunchKernelGGL((attn_fwd_kernel<48, true, 4>), grid, dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<48, false, 4>), grid, dim3(256), 0, (hipStream_t)stream_, a);
  }
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" size_t advmil_mha_bwd_workspace_bytes(int64_t Ltot, int nhead) { return (size_t)Ltot * (size_t)nhead * sizeof(float); }

extern "C" int advmil_mha_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int64_t Ltot, int nhead,
                              int head_dim, int nseg, const int64_t* ptr, int64_t max_len, float drop_p, const uint64_t* seed,
                              uint64_t stream_id, const int64_t* rng_rowoff, float* dqkv, void* ws, size_t ws_bytes,
                              advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  AttnArgs a;
  int nw = 4;
  const int rc = attn_args(a, qkv, Ltot, nhead, head_dim, nseg, ptr, max_len, drop_p, seed, stream_id, rng_rowoff, nw);
  if (rc) return rc;
  if (!out || !dout || !lse || !dqkv || !ws) return ADVMIL_EINVAL;
  if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15) || ((uintptr_t)ws & 15)) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_mha_bwd_workspace_bytes(Ltot, nhead)) return ADVMIL_EWORKSPACE;
  float* dsum = (float*)ws;
  const int64_t n = Ltot * nhead;
  hipLaunchKernelGGL((attn_bwd_prep_kernel<48>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dout, out, n, dsum);
  ADVMIL_LAUNCH_CHECK();
  a.lse = const_cast<float*>(lse); a.dout = dout; a.dsum = dsum; a.dqkv = dqkv;
  const dim3 grid((unsigned)(a.ntile * nseg * nhead));
  if (nw == 8) {
    if (a.seed) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<48, true, 8>), grid, dim3(512), 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<48, true, 8>), grid, dim3(512), 0, stream, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<48, false, 8>), grid, dim3(512), 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<48, false, 8>), grid, dim3(512), 0, stream, a);
    }
  } else if (a.seed) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<48, true, 4>), grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<48, true, 4>), grid, dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<48, false, 4>), grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<48, false, 4>), grid, dim3(256), 0, stream, a);
  }
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
