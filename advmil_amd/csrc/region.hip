// The discriminator's region-level network as ONE launch each way (reference model/model_utils.py:188-210 EmbedXLayer after the region
// embedding: fc1 = Linear(d, d/2) -> ReLU -> Dropout -> Linear(d/2, d), then GAPool's scorer, model/backbone_utils.py:31-56:
// tanh(Linear(d, d)) * sigmoid(Linear(d, d)) -> Linear(d, 1)), d = 128 -- rows = the regions of a step slab (N / 16 per bag):
//     h1 = dropout(relu(e W1^T + b1))   [R, 64]
//     fc = h1 W2^T + b2                 [R, 128]      (the tensor that is pooled, and region-averaged for the RLIP inner product)
//     a | b = tanh | sigmoid (fc Wab^T + bab)         [R, 256]
//     s = sum_j drop(a_j) drop(b_j) wc_j + bc         [R]
// As separate launches these are three 64x64-tile contractions + a gate-score pass forward (8-21 us each at 16 384 rows, their 5-10 us
// latency floor at the 2 048 rows of a two-bag step) and ten launches backward. Here a workgroup owns 64 rows: every layer's input
// tile sits in LDS as bf16x3 operand planes (hi = bf16(x), lo = bf16(x - hi), 16-byte units XOR-swizzled for conflict-free ds_read_b128
// fragments), the weights' planes (kept current by the Adam kernel) are read as B fragments straight from L2 (192 KB per tile), products
// on v_mfma_f32_32x32x16_bf16 in the contraction engine's order (al.bh + ah.bl + ah.bh, fp32 accumulate), every epilogue through a
// wave-private fp32 patch so that global stores are 16 B per lane along the row. bf16x3 arithmetic only (gemm_mode 'exact' keeps the
// layer-by-layer path). The backward kernel produces dG | d fc | d pre1 | d e per row; the three weight gradients stay deep-K
// contractions of the engine over those rows.
#include "common.h"
#include "bf16split.h"
#include "sumq.h"
#include "../../include/advmil_hip.h"

#define RC_D 128
#define RC_H 64
#define RC_ROWS 64
#define RC_PITCH 36      // fp32 patch row pitch (32 + 4)

// Workgroup barrier for LDS data only: __syncthreads() also drains vmcnt, i.e. waits for the ACKs of every global store issued so far
// (the saved activations of the layer just finished: 1-2 us each time); everything these kernels exchange between waves is in LDS.
#define LDS_BARRIER()                                   \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
  } while (0)

#define RC_WAVE_SYNC()                                     \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_s_waitcnt(0xc07f);                    \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// halfword offset of 16-byte unit `unit` of row `row` in a [rows][K] bf16 plane image. Row pitch 128 B (K = 64): two rows share a
// 256-byte bank row -> XOR with (row >> 1) & 7; pitch 256 / 512 B: XOR with row & 15. With these a ds_read_b128 lane group
// ({0-3, 12-15, 20-27} and its shifts: rows distinct mod 16) covers 16 distinct 16-byte slots.
template <int K>
__device__ __forceinline__ int rc_off(int row, int unit) {
  const int u = (K == 64) ? (unit ^ ((row >> 1) & 7)) : (unit ^ (row & 15));
  return row * K + u * 8;
}

// acc[a][b] += A[rows of block rb0 + a] . B[cols n0[b] ..]^T over K: A = LDS plane images (hi, lo) [64][K], B = global planes [n][K]
// The weight fragments of a WHOLE product are requested up front (K / 16 x NCB x 2 planes x 4 registers: 32-128 VGPRs; one workgroup per
// CU leaves a wave the full register file) -- and, in the kernels below, one product AHEAD: the next layer's fragments are in flight
// while the current layer's epilogue runs, the first layers' (and every bias / scorer vector) from the top of the kernel. Per launch
// that leaves ~3 dependent global round trips instead of ~8 (a 64-row tile is latency, not work: 23-27 us per launch whatever the rows).
template <int K, int NCB>
struct RcFrags {
  Frag8 h[K / 16][NCB], l[K / 16][NCB];
  __device__ __forceinline__ void load(const bf16raw* __restrict__ Bh, const bf16raw* __restrict__ Bl, const int (&n0)[NCB], int lane) {
    const int i = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks)
#pragma unroll
      for (int b = 0; b < NCB; ++b) {
        const int64_t o = (int64_t)(n0[b] + i) * K + ks * 16 + hi * 8;
        h[ks][b].u = *reinterpret_cast<const uint4*>(Bh + o);
        l[ks][b].u = *reinterpret_cast<const uint4*>(Bl + o);
      }
    __builtin_amdgcn_sched_barrier(0);       // (keeps the scheduler from sinking the loads back to their first use)
  }
};

// acc[a][b] += A[rows of block rb0 + a] . B[cols of fragment set b]^T over K: A = LDS plane images (hi, lo) [64][K], B = fragments in registers
template <int K, int NRB, int NCB>
__device__ __forceinline__ void rc_mma(const bf16raw* __restrict__ Ah, const bf16raw* __restrict__ Al, int rb0, const RcFrags<K, NCB>& f,
                                       f32x16 (&acc)[NRB][NCB], int lane) {
  const int i = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < K / 16; ++ks) {
    Frag8 ah[NRB], al[NRB];
#pragma unroll
    for (int a = 0; a < NRB; ++a) {
      const int o = rc_off<K>((rb0 + a) * 32 + i, ks * 2 + hi);
      ah[a].u = *reinterpret_cast<const uint4*>(Ah + o);
      al[a].u = *reinterpret_cast<const uint4*>(Al + o);
    }
#pragma unroll
    for (int a = 0; a < NRB; ++a)
#pragma unroll
      for (int b = 0; b < NCB; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a].v, f.h[ks][b].v, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a].v, f.l[ks][b].v, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a].v, f.h[ks][b].v, acc[a][b], 0, 0, 0);
      }
  }
}

// accumulator block (MFMA C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) -> the wave's fp32 patch
__device__ __forceinline__ void rc_to_patch(const f32x16& acc, float* __restrict__ patch, int lane) {
  const int i = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * RC_PITCH + i] = acc[r];
}
// lane's four consecutive columns c4 = (lane & 7) * 4 of patch row q * 8 + (lane >> 3)
__device__ __forceinline__ float4 rc_from_patch(const float* __restrict__ patch, int q, int lane) {
  return *reinterpret_cast<const float4*>(patch + (q * 8 + (lane >> 3)) * RC_PITCH + (lane & 7) * 4);
}
// 4 consecutive values of row `row`, columns col .. col + 3 -> the [64][K] plane images
template <int K>
__device__ __forceinline__ void rc_put_planes(bf16raw* __restrict__ Ph, bf16raw* __restrict__ Pl, int row, int col, const float4& v) {
  uint2 h, l;
  split4(v, h, l);
  const int o = rc_off<K>(row, col >> 3) + (col & 4);
  *reinterpret_cast<uint2*>(Ph + o) = h;
  *reinterpret_cast<uint2*>(Pl + o) = l;
}

struct RcFwdArgs {
  const float* e;
  int64_t R;
  const bf16raw *W1h, *W1l, *W2h, *W2l, *Wabh, *Wabl;
  const float *b1, *b2, *bab, *wc, *bc;
  float p1, pg;
  const uint64_t* seed;
  uint64_t sid1, sida, sidb;
  const int64_t* rng_row;
  float *h1, *fc, *ab, *s;
};

__global__ __launch_bounds__(256) void dx_chain_fwd_kernel(RcFwdArgs g) {
  __shared__ __attribute__((aligned(16))) bf16raw sX[2 * RC_ROWS * RC_D];      // planes of e, then of fc (hi | lo)
  __shared__ __attribute__((aligned(16))) bf16raw sH[2 * RC_ROWS * RC_H];      // planes of h1
  __shared__ __attribute__((aligned(16))) float sP[4][2][32 * RC_PITCH];
  __shared__ float sS[4][RC_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * RC_ROWS;
  bf16raw* const Xh = sX; bf16raw* const Xl = sX + RC_ROWS * RC_D;
  bf16raw* const Hh = sH; bf16raw* const Hl = sH + RC_ROWS * RC_H;
  // every parameter read of the launch that does not depend on the rows, up front: the first two layers' weight fragments, the biases
  // and the scorer's output weights of this lane's columns
  const int c4 = (lane & 7) * 4, rq = lane >> 3;
  const int n1[1] = {(wave & 1) * 32}, n2[1] = {wave * 32}, ng[2] = {wave * 32, RC_D + wave * 32};
  RcFrags<RC_D, 1> f1;
  RcFrags<RC_H, 1> f2;
  f1.load(g.W1h, g.W1l, n1, lane);
  f2.load(g.W2h, g.W2l, n2, lane);
  const float4 b1v = *reinterpret_cast<const float4*>(g.b1 + n1[0] + c4);
  const float4 b2v = *reinterpret_cast<const float4*>(g.b2 + n2[0] + c4);
  const float4 bav = *reinterpret_cast<const float4*>(g.bab + wave * 32 + c4);
  const float4 bbv = *reinterpret_cast<const float4*>(g.bab + RC_D + wave * 32 + c4);
  const float4 wv = *reinterpret_cast<const float4*>(g.wc + wave * 32 + c4);
  const float bcv = g.bc ? g.bc[0] : 0.f;
  for (int idx = tid; idx < RC_ROWS * (RC_D / 4); idx += 256) {
    const int row = idx / (RC_D / 4), c4e = idx % (RC_D / 4);
    const float4 v = (r0 + row < g.R) ? *reinterpret_cast<const float4*>(g.e + (r0 + row) * RC_D + c4e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    rc_put_planes<RC_D>(Xh, Xl, row, c4e * 4, v);
  }
  LDS_BARRIER();
  const bool d1 = g.seed && g.p1 > 0.f, dg = g.seed && g.pg > 0.f;
  uint64_t k1 = 0, ka = 0, kb = 0;
  float inv1 = 1.f, invg = 1.f;
  if (d1 || dg) {
    const uint64_t sd = *g.seed;
    if (d1) { k1 = rng_key(sd, g.sid1); inv1 = hw_rcp(1.f - g.p1); }
    if (dg) { ka = rng_key(sd, g.sida); kb = rng_key(sd, g.sidb); invg = hw_rcp(1.f - g.pg); }
  }
  float* const pa = sP[wave][0];
  float* const pb = sP[wave][1];
  // ---- layer 1: wave -> block (rb = wave >> 1, columns 32 (wave & 1) ..)
  {
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int rb = wave >> 1;
    rc_mma<RC_D, 1, 1>(Xh, Xl, rb, f1, acc, lane);
    rc_to_patch(acc[0][0], pa, lane);
    RC_WAVE_SYNC();
    const int col = n1[0] + c4;
    const float4 bv = b1v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rb * 32 + q * 8 + rq;
      float4 v = rc_from_patch(pa, q, lane);
      v.x = fmaxf(v.x + bv.x, 0.f); v.y = fmaxf(v.y + bv.y, 0.f); v.z = fmaxf(v.z + bv.z, 0.f); v.w = fmaxf(v.w + bv.w, 0.f);
      const bool ok = r0 + row < g.R;
      if (d1) {
        const int64_t grow = (ok && g.rng_row) ? g.rng_row[r0 + row] : r0 + row;
        const uint64_t ix = (uint64_t)(grow * RC_H + col);
        v.x *= rng_keep(k1, ix, g.p1, inv1); v.y *= rng_keep(k1, ix + 1, g.p1, inv1);
        v.z *= rng_keep(k1, ix + 2, g.p1, inv1); v.w *= rng_keep(k1, ix + 3, g.p1, inv1);
      }
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && g.h1) *reinterpret_cast<float4*>(g.h1 + (r0 + row) * RC_H + col) = v;
      rc_put_planes<RC_H>(Hh, Hl, row, col, v);
    }
  }
  RcFrags<RC_D, 2> fg;          // the scorer's fragments: in flight under layer 2
  fg.load(g.Wabh, g.Wabl, ng, lane);
  LDS_BARRIER();          // h1's planes complete; every wave is done reading e's planes
  // ---- layer 2: wave -> columns 32 wave .., both row blocks
  {
    f32x16 acc[2][1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
    rc_mma<RC_H, 2, 1>(Hh, Hl, 0, f2, acc, lane);
    const int col = n2[0] + c4;
    const float4 bv = b2v;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      rc_to_patch(acc[a][0], pa, lane);
      RC_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = a * 32 + q * 8 + rq;
        float4 v = rc_from_patch(pa, q, lane);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        const bool ok = r0 + row < g.R;
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) *reinterpret_cast<float4*>(g.fc + (r0 + row) * RC_D + col) = v;
        rc_put_planes<RC_D>(Xh, Xl, row, col, v);
      }
      RC_WAVE_SYNC();
    }
  }
  LDS_BARRIER();          // fc's planes complete
  // ---- gates: wave -> tanh columns 32 wave .. and sigmoid columns 128 + 32 wave .., both row blocks
  {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    rc_mma<RC_D, 2, 2>(Xh, Xl, 0, fg, acc, lane);
    const int col = wave * 32 + c4;            // j of the lane's four gate units
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      rc_to_patch(acc[a][0], pa, lane);
      rc_to_patch(acc[a][1], pb, lane);
      RC_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = a * 32 + q * 8 + rq;
        const float4 x = rc_from_patch(pa, q, lane), y = rc_from_patch(pb, q, lane);
        float av[4] = {act_apply(ACT_TANH, x.x + bav.x), act_apply(ACT_TANH, x.y + bav.y), act_apply(ACT_TANH, x.z + bav.z), act_apply(ACT_TANH, x.w + bav.w)};
        float bv[4] = {act_apply(ACT_SIGMOID, y.x + bbv.x), act_apply(ACT_SIGMOID, y.y + bbv.y), act_apply(ACT_SIGMOID, y.z + bbv.z),
                       act_apply(ACT_SIGMOID, y.w + bbv.w)};
        const bool ok = r0 + row < g.R;
        if (ok && g.ab) {
          *reinterpret_cast<float4*>(g.ab + (r0 + row) * 2 * RC_D + col) = make_float4(av[0], av[1], av[2], av[3]);
          *reinterpret_cast<float4*>(g.ab + (r0 + row) * 2 * RC_D + RC_D + col) = make_float4(bv[0], bv[1], bv[2], bv[3]);
        }
        const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
        float t = 0.f;
        if (dg) {
          const int64_t grow = (ok && g.rng_row) ? g.rng_row[r0 + row] : r0 + row;
          const uint64_t ix = (uint64_t)(grow * RC_D + col);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            t += (av[u] * rng_keep(ka, ix + u, g.pg, invg)) * (bv[u] * rng_keep(kb, ix + u, g.pg, invg)) * wq[u];
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) t += av[u] * bv[u] * wq[u];
        }
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);      // the 8 lanes that share a row
        if ((lane & 7) == 0) sS[wave][row] = t;
      }
      RC_WAVE_SYNC();
    }
  }
  LDS_BARRIER();
  if (tid < RC_ROWS && r0 + tid < g.R) g.s[r0 + tid] = ((sS[0][tid] + sS[1][tid]) + (sS[2][tid] + sS[3][tid])) + bcv;
}

// ======================================================================================================================================
// backward, per 64-row tile:
//   dG   = gate backward of ds (tanh', sigmoid', the two dropout draws recomputed)             [64, 256]  -> global + LDS planes
//   dfc  = dG WabT^T + A[r] dpooled[bag(r)] + dmean[bag(r)] / len(bag(r)) (+ dfc_add[r])       [64, 128]  -> global + LDS planes
//   dpre = (dfc W2T^T) * (h1 > 0 ? 1 / (1 - p1) : 0)                                           [64, 64]   -> global + LDS planes
//   de   = dpre W1T^T                                                                          [64, 128]  -> global
// WabT / W2T / W1T: the weights TRANSPOSED, as operand planes (advmil_dx_chain_prep). Column sums of the tile (dwc | dba | dbb | dbc |
// db2 | db1) go to one partial row per workgroup; the host side merges them (sumq) into the gradient arena.
// ======================================================================================================================================
#define RC_PART (3 * RC_D + 4 + RC_D + RC_H)      // dwc 128 | dba 128 | dbb 128 | dbc 1 + 3 pad | db2 128 | db1 64

struct RcBwdArgs {
  int64_t R;
  const float *ds, *A, *dpooled, *dmean, *dfc_add;
  const int32_t* rowseg;
  const int64_t* segptr;
  const float *h1, *ab, *wc;
  float p1, pg;
  const uint64_t* seed;
  uint64_t sida, sidb;
  const int64_t* rng_row;
  const bf16raw *WabTh, *WabTl, *W2Th, *W2Tl, *W1Th, *W1Tl;
  float *dG, *dfc, *dpre, *de, *partial;
};

__global__ __launch_bounds__(256) void dx_chain_bwd_kernel(RcBwdArgs g) {
  __shared__ __attribute__((aligned(16))) bf16raw sG[2 * RC_ROWS * 2 * RC_D];   // planes of dG [64][256]
  __shared__ __attribute__((aligned(16))) bf16raw sF[2 * RC_ROWS * RC_D];       // planes of dfc
  __shared__ __attribute__((aligned(16))) bf16raw sD[2 * RC_ROWS * RC_H];       // planes of dpre
  __shared__ __attribute__((aligned(16))) float sP[4][32 * RC_PITCH];
  __shared__ __attribute__((aligned(16))) float sR[8][3 * RC_D];                // column-sum staging of the gate backward
  __shared__ float sB1[2][RC_H];
  __shared__ float sDs[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * RC_ROWS;
  bf16raw* const Gh = sG; bf16raw* const Gl = sG + RC_ROWS * 2 * RC_D;
  bf16raw* const Fh = sF; bf16raw* const Fl = sF + RC_ROWS * RC_D;
  bf16raw* const Dh = sD; bf16raw* const Dl = sD + RC_ROWS * RC_H;
  float* const prow = g.partial + (int64_t)blockIdx.x * RC_PART;
  const bool dg = g.seed && g.pg > 0.f;
  uint64_t ka = 0, kb = 0;
  float invg = 1.f;
  if (dg) { const uint64_t sd = *g.seed; ka = rng_key(sd, g.sida); kb = rng_key(sd, g.sidb); invg = hw_rcp(1.f - g.pg); }
  // the first product's weight fragments (WabT, 128 registers) fly under the gate backward
  const int nA[1] = {wave * 32}, nB[1] = {(wave & 1) * 32};
  RcFrags<2 * RC_D, 1> fA;
  fA.load(g.WabTh, g.WabTl, nA, lane);
  // ---- gate backward: thread -> 4 gate units j = 4 (tid & 31) .., rows (tid >> 5) + 8 it
  {
    const int j = (tid & 31) * 4, rg = tid >> 5;
    const float4 w4 = *reinterpret_cast<const float4*>(g.wc + j);
    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
    float swc[4] = {0.f, 0.f, 0.f, 0.f}, sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    float sds = 0.f;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
      const int row = rg + 8 * it;
      const int64_t n = r0 + row;
      float ga[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
      if (n < g.R) {
        const float d = g.ds[n];
        if ((tid & 31) == 0) sds += d;
        const float4 a4 = *reinterpret_cast<const float4*>(g.ab + n * 2 * RC_D + j);
        const float4 b4 = *reinterpret_cast<const float4*>(g.ab + n * 2 * RC_D + RC_D + j);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
        const uint64_t ix = (uint64_t)((g.rng_row ? g.rng_row[n] : n) * RC_D + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float fa = 1.f, fb = 1.f;
          if (dg) { fa = rng_keep(ka, ix + q, g.pg, invg); fb = rng_keep(kb, ix + q, g.pg, invg); }
          const float ad = av[q] * fa, bd = bv[q] * fb;
          swc[q] += d * ad * bd;
          ga[q] = d * wv[q] * bd * fa * (1.f - av[q] * av[q]);
          gb[q] = d * wv[q] * ad * fb * bv[q] * (1.f - bv[q]);
          sa[q] += ga[q]; sb[q] += gb[q];
        }
        *reinterpret_cast<float4*>(g.dG + n * 2 * RC_D + j) = make_float4(ga[0], ga[1], ga[2], ga[3]);
        *reinterpret_cast<float4*>(g.dG + n * 2 * RC_D + RC_D + j) = make_float4(gb[0], gb[1], gb[2], gb[3]);
      }
      rc_put_planes<2 * RC_D>(Gh, Gl, row, j, make_float4(ga[0], ga[1], ga[2], ga[3]));
      rc_put_planes<2 * RC_D>(Gh, Gl, row, RC_D + j, make_float4(gb[0], gb[1], gb[2], gb[3]));
    }
    *reinterpret_cast<float4*>(&sR[rg][j]) = make_float4(swc[0], swc[1], swc[2], swc[3]);
    *reinterpret_cast<float4*>(&sR[rg][RC_D + j]) = make_float4(sa[0], sa[1], sa[2], sa[3]);
    *reinterpret_cast<float4*>(&sR[rg][2 * RC_D + j]) = make_float4(sb[0], sb[1], sb[2], sb[3]);
    if ((tid & 31) == 0) sDs[rg] = sds;
  }
  LDS_BARRIER();          // dG's planes and the column-sum staging complete
  for (int c = tid; c < 3 * RC_D; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sR[r][c];
    prow[c] = t;
  }
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sDs[r];
    prow[3 * RC_D] = t;
  }
  const int c4 = (lane & 7) * 4, rq = lane >> 3;
  float* const pa = sP[wave];
  // ---- dfc = dG WabT^T (+ the pooling's direct paths): wave -> columns 32 wave .., both row blocks
  {
    f32x16 acc[2][1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
    rc_mma<2 * RC_D, 2, 1>(Gh, Gl, 0, fA, acc, lane);
    const int col = nA[0] + c4;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      rc_to_patch(acc[a][0], pa, lane);
      RC_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = a * 32 + q * 8 + rq;
        const int64_t n = r0 + row;
        float4 v = rc_from_patch(pa, q, lane);
        const bool ok = n < g.R;
        if (ok) {
          const int sg = g.rowseg ? g.rowseg[n] : 0;
          const float An = g.A[n];
          const float4 dp = *reinterpret_cast<const float4*>(g.dpooled + (int64_t)sg * RC_D + col);
          v.x += An * dp.x; v.y += An * dp.y; v.z += An * dp.z; v.w += An * dp.w;
          if (g.dmean) {
            const float il = 1.0f / (float)(g.segptr ? (g.segptr[sg + 1] - g.segptr[sg]) : g.R);
            const float4 dm = *reinterpret_cast<const float4*>(g.dmean + (int64_t)sg * RC_D + col);
            v.x += il * dm.x; v.y += il * dm.y; v.z += il * dm.z; v.w += il * dm.w;
          }
          if (g.dfc_add) {
            const float4 x = *reinterpret_cast<const float4*>(g.dfc_add + n * RC_D + col);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
          }
          *reinterpret_cast<float4*>(g.dfc + n * RC_D + col) = v;
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
        rc_put_planes<RC_D>(Fh, Fl, row, col, v);
      }
      RC_WAVE_SYNC();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cs[u] += __shfl_xor(cs[u], 8, 64); cs[u] += __shfl_xor(cs[u], 16, 64); cs[u] += __shfl_xor(cs[u], 32, 64);
    }
    if (rq == 0) *reinterpret_cast<float4*>(prow + 3 * RC_D + 4 + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);      // db2
  }
  // the two remaining products' fragments (W2T 64 + W1T 32 registers): requested before the barrier, consumed behind it
  RcFrags<RC_D, 1> fB;
  RcFrags<RC_H, 1> fC;
  fB.load(g.W2Th, g.W2Tl, nB, lane);
  if (g.de) fC.load(g.W1Th, g.W1Tl, nA, lane);
  LDS_BARRIER();          // dfc's planes complete
  // ---- dpre = (dfc W2T^T) masked by h1: wave -> block (rb = wave >> 1, columns 32 (wave & 1) ..)
  {
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int rb = wave >> 1;
    rc_mma<RC_D, 1, 1>(Fh, Fl, rb, fB, acc, lane);
    rc_to_patch(acc[0][0], pa, lane);
    RC_WAVE_SYNC();
    const int col = nB[0] + c4;
    const float inv1 = g.p1 > 0.f ? hw_rcp(1.f - g.p1) : 1.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rb * 32 + q * 8 + rq;
      const int64_t n = r0 + row;
      float4 v = rc_from_patch(pa, q, lane);
      if (n < g.R) {
        const float4 h = *reinterpret_cast<const float4*>(g.h1 + n * RC_H + col);
        v.x = h.x > 0.f ? v.x * inv1 : 0.f; v.y = h.y > 0.f ? v.y * inv1 : 0.f;
        v.z = h.z > 0.f ? v.z * inv1 : 0.f; v.w = h.w > 0.f ? v.w * inv1 : 0.f;
        *reinterpret_cast<float4*>(g.dpre + n * RC_H + col) = v;
      } else {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
      rc_put_planes<RC_H>(Dh, Dl, row, col, v);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cs[u] += __shfl_xor(cs[u], 8, 64); cs[u] += __shfl_xor(cs[u], 16, 64); cs[u] += __shfl_xor(cs[u], 32, 64);
    }
    if (rq == 0) *reinterpret_cast<float4*>(&sB1[rb][col]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
  }
  LDS_BARRIER();          // dpre's planes and the two row blocks' db1 pieces complete
  if (tid < RC_H) prow[3 * RC_D + 4 + RC_D + tid] = sB1[0][tid] + sB1[1][tid];
  if (g.de) {
    // ---- de = dpre W1T^T: wave -> columns 32 wave .., both row blocks
    f32x16 acc[2][1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
    rc_mma<RC_H, 2, 1>(Dh, Dl, 0, fC, acc, lane);
    const int col = nA[0] + c4;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      rc_to_patch(acc[a][0], pa, lane);
      RC_WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t n = r0 + a * 32 + q * 8 + rq;
        if (n < g.R) *reinterpret_cast<float4*>(g.de + n * RC_D + col) = rc_from_patch(pa, q, lane);
      }
      RC_WAVE_SYNC();
    }
  }
}

// transposed operand planes of the three weights for the backward: dst[k][n] = src[n][k] split into hi / lo
__global__ __launch_bounds__(256) void dx_chain_prep_kernel(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ Wab,
                                                            bf16raw* __restrict__ W1Th, bf16raw* __restrict__ W1Tl, bf16raw* __restrict__ W2Th,
                                                            bf16raw* __restrict__ W2Tl, bf16raw* __restrict__ WabTh, bf16raw* __restrict__ WabTl) {
  const int total = RC_H * RC_D * 2 + 2 * RC_D * RC_D;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < total; o += gridDim.x * 256) {
    const float* src; bf16raw *dh, *dl; int N, K, e;
    if (o < RC_H * RC_D) { src = W1; dh = W1Th; dl = W1Tl; N = RC_H; K = RC_D; e = o; }                           // W1 [64][128] -> [128][64]
    else if (o < 2 * RC_H * RC_D) { src = W2; dh = W2Th; dl = W2Tl; N = RC_D; K = RC_H; e = o - RC_H * RC_D; }     // W2 [128][64] -> [64][128]
    else { src = Wab; dh = WabTh; dl = WabTl; N = 2 * RC_D; K = RC_D; e = o - 2 * RC_H * RC_D; }                   // Wab [256][128] -> [128][256]
    const int k = e / N, n = e - k * N;            // destination element [k][n] (n contiguous: coalesced stores)
    const float v = src[(int64_t)n * K + k];
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    dh[e] = *reinterpret_cast<const bf16raw*>(&h);
    dl[e] = *reinterpret_cast<const bf16raw*>(&l);
  }
}

extern "C" int advmil_dx_chain_fwd(const float* e, int64_t R, int d, const void* W1_hi, const void* W1_lo, const float* b1, const void* W2_hi,
                                   const void* W2_lo, const float* b2, const void* Wab_hi, const void* Wab_lo, const float* bab, const float* wc,
                                   const float* bc, float p1, float pg, const uint64_t* seed, uint64_t sid1, uint64_t sida, uint64_t sidb,
                                   const int64_t* rng_row, float* h1, float* fc, float* ab, float* s, advmil_stream_t stream_) {
  if (!e || R <= 0 || d != RC_D || !W1_hi || !W1_lo || !W2_hi || !W2_lo || !Wab_hi || !Wab_lo || !b1 || !b2 || !bab || !wc || !fc || !s)
    return ADVMIL_EINVAL;
  if (!(p1 >= 0.f && p1 < 1.f) || !(pg >= 0.f && pg < 1.f)) return ADVMIL_EINVAL;
  if ((((uintptr_t)e) | ((uintptr_t)W1_hi) | ((uintptr_t)W1_lo) | ((uintptr_t)W2_hi) | ((uintptr_t)W2_lo) | ((uintptr_t)Wab_hi) | ((uintptr_t)Wab_lo) |
       ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)bab) | ((uintptr_t)wc) | ((uintptr_t)h1) | ((uintptr_t)fc) | ((uintptr_t)ab)) & 15)
    return ADVMIL_EINVAL;
  RcFwdArgs g;
  g.e = e; g.R = R;
  g.W1h = (const bf16raw*)W1_hi; g.W1l = (const bf16raw*)W1_lo; g.W2h = (const bf16raw*)W2_hi; g.W2l = (const bf16raw*)W2_lo;
  g.Wabh = (const bf16raw*)Wab_hi; g.Wabl = (const bf16raw*)Wab_lo;
  g.b1 = b1; g.b2 = b2; g.bab = bab; g.wc = wc; g.bc = bc;
  g.p1 = p1; g.pg = pg; g.seed = seed; g.sid1 = sid1; g.sida = sida; g.sidb = sidb; g.rng_row = rng_row;
  g.h1 = h1; g.fc = fc; g.ab = ab; g.s = s;
  hipLaunchKernelGGL(dx_chain_fwd_kernel, dim3((unsigned)((R + RC_ROWS - 1) / RC_ROWS)), dim3(256), 0, (hipStream_t)stream_, g);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" size_t advmil_dx_chain_bwd_workspace_bytes(int64_t R, int d) {
  (void)d;
  return (size_t)((R + RC_ROWS - 1) / RC_ROWS) * RC_PART * sizeof(float);
}

extern "C" int advmil_dx_chain_prep(const float* W1, const float* W2, const float* Wab, int d, void* W1T_hi, void* W1T_lo, void* W2T_hi,
                                    void* W2T_lo, void* WabT_hi, void* WabT_lo, advmil_stream_t stream_) {
  if (!W1 || !W2 || !Wab || d != RC_D || !W1T_hi || !W1T_lo || !W2T_hi || !W2T_lo || !WabT_hi || !WabT_lo) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(dx_chain_prep_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream_, W1, W2, Wab, (bf16raw*)W1T_hi, (bf16raw*)W1T_lo,
                     (bf16raw*)W2T_hi, (bf16raw*)W2T_lo, (bf16raw*)WabT_hi, (bf16raw*)WabT_lo);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_dx_chain_bwd(int64_t R, int d, const float* ds, const float* A, const float* dpooled, const float* dmean,
                                   const int32_t* rowseg, const int64_t* seg_ptr, const float* dfc_add, const float* h1, const float* ab,
                                   const float* wc, float p1,
                                   float pg, const uint64_t* seed, uint64_t sida, uint64_t sidb, const int64_t* rng_row, const void* WabT_hi,
                                   const void* WabT_lo, const void* W2T_hi, const void* W2T_lo, const void* W1T_hi, const void* W1T_lo, float* dG,
                                   float* dfc, float* dpre, float* de, float* dwc, float* dbab, float* dbc, float* db2, float* db1, void* ws,
                                   size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (R <= 0 || d != RC_D || !ds || !A || !dpooled || !h1 || !ab || !wc || !WabT_hi || !WabT_lo || !W2T_hi || !W2T_lo || !dG || !dfc || !dpre ||
      !dwc || !dbab || !dbc || !db2 || !db1 || !ws)
    return ADVMIL_EINVAL;
  if (de && (!W1T_hi || !W1T_lo)) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_dx_chain_bwd_workspace_bytes(R, d)) return ADVMIL_EWORKSPACE;
  if (!(p1 >= 0.f && p1 < 1.f) || !(pg >= 0.f && pg < 1.f)) return ADVMIL_EINVAL;
  if ((((uintptr_t)dpooled) | ((uintptr_t)dmean) | ((uintptr_t)dfc_add) | ((uintptr_t)h1) | ((uintptr_t)ab) | ((uintptr_t)wc) | ((uintptr_t)WabT_hi) | ((uintptr_t)WabT_lo) |
       ((uintptr_t)W2T_hi) | ((uintptr_t)W2T_lo) | ((uintptr_t)W1T_hi) | ((uintptr_t)W1T_lo) | ((uintptr_t)dG) | ((uintptr_t)dfc) | ((uintptr_t)dpre) |
       ((uintptr_t)de) | ((uintptr_t)ws)) & 15)
    return ADVMIL_EINVAL;
  RcBwdArgs g;
  g.R = R; g.ds = ds; g.A = A; g.dpooled = dpooled; g.dmean = dmean; g.dfc_add = dfc_add; g.rowseg = rowseg; g.segptr = seg_ptr;
  g.h1 = h1; g.ab = ab; g.wc = wc; g.p1 = p1; g.pg = pg; g.seed = seed; g.sida = sida; g.sidb = sidb; g.rng_row = rng_row;
  g.WabTh = (const bf16raw*)WabT_hi; g.WabTl = (const bf16raw*)WabT_lo; g.W2Th = (const bf16raw*)W2T_hi; g.W2Tl = (const bf16raw*)W2T_lo;
  g.W1Th = (const bf16raw*)W1T_hi; g.W1Tl = (const bf16raw*)W1T_lo;
  g.dG = dG; g.dfc = dfc; g.dpre = dpre; g.de = de; g.partial = (float*)ws;
  const int nblk = (int)((R + RC_ROWS - 1) / RC_ROWS);
  hipLaunchKernelGGL(dx_chain_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, g);
  ADVMIL_LAUNCH_CHECK();
  // the per-workgroup partial rows -> the (accumulating) parameter gradients; merged at once, or with the backward's other partial sums
  // when the stream is in deferral (sumq.hip)
  int rc = advmil_sumq(stream, (const float*)ws, nblk, RC_PART, 3 * RC_D + 1, dwc, 1, dbab, RC_D, dbc, 3 * RC_D);
  if (!rc) rc = advmil_sumq(stream, (const float*)ws + 3 * RC_D + 4, nblk, RC_PART, RC_D + RC_H, db2, 1, db1, RC_D);
  return rc;
}
