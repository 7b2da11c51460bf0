// HBM-bound kernels of the AdvMIL path: gated-attention score, instance softmax + weighted pooling,
// their backward, the Linear activation/dropout backward, the LayerNorm-ReLU-mean16 region embedding
// tail, and column-sum utilities. All loads are 16 B/lane (float4) along the contiguous dimension or
// one-wave-per-row with 256 B coalesced segments; reductions are deterministic two-stage (per-workgroup
// partials in a caller workspace, then a merge launch) -- no float atomics.
#include <cstdlib>
#include "common.h"
#include "bf16split.h"
#include "sumq.h"
#include "../../include/advmil_hip.h"

#define ROWS_PER_BLOCK 32
// rows per workgroup of the two-stage reductions: 32 for one bag, grown so that a step slab yields <= ~1024 partial rows
// (the merge kernels walk the partials serially per column)
static inline int rows_per_block(int64_t rows) {
  int64_t r = (rows + 1023) / 1024;
  r = (r + 31) / 32 * 32;
  if (r < ROWS_PER_BLOCK) r = ROWS_PER_BLOCK;
  if (r > 512) r = 512;
  return (int)r;
}

// 256 threads cover `rpp` rows x `cols4` float4 columns
struct RowColMap {
  int cols4, rpp, c4, r;
  bool active;
};
__device__ __forceinline__ RowColMap make_map(int64_t D) {
  RowColMap m;
  m.cols4 = (int)(D >> 2);
  m.rpp = 256 / m.cols4;
  if (m.rpp < 1) m.rpp = 1;
  m.c4 = threadIdx.x % m.cols4;
  m.r = threadIdx.x / m.cols4;
  m.active = m.r < m.rpp;
  return m;
}

// sum a float4 held by each (r, c4) thread over r; result valid in threads with r == 0
__device__ __forceinline__ float4 reduce_rows(float4 v, const RowColMap& m, float* red /* >= 1024 floats */) {
  __syncthreads();
  if (m.active) *reinterpret_cast<float4*>(red + (m.r * m.cols4 + m.c4) * 4) = v;
  __syncthreads();
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.active && m.r == 0) {
    for (int r = 0; r < m.rpp; ++r) {
      const float4 p = *reinterpret_cast<const float4*>(red + (r * m.cols4 + m.c4) * 4);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
  }
  return s;
}

// out[c] = sum_b partial[b*stride + c].  16 columns x 16 row-lanes per workgroup: every lane keeps 16 independent
// loads in flight and the 16 row-lanes are folded through LDS, so the serial depth is nblk/256 (was nblk).
__global__ __launch_bounds__(256) void colsum_merge_kernel(const float* __restrict__ partial, int nblk, int64_t stride,
                                                           int64_t ncols, float* __restrict__ out, int accumulate,
                                                           int64_t seg_pstride = 0, int64_t seg_ostride = 0,
                                                           float* __restrict__ out1 = nullptr, int64_t c1 = 0,
                                                           float* __restrict__ out2 = nullptr, int64_t c2 = 0) {
  __shared__ float red[16][17];
  partial += (int64_t)blockIdx.y * seg_pstride;      // gridDim.y = segments (bags of a step slab)
  out += (int64_t)blockIdx.y * seg_ostride;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 16 + cl;
  // 16 independent loads in flight per lane: the walk is latency-bound (a slab leaves up to 1024 partial rows, i.e. 64 per
  // row-lane; with 4 in flight the merge took 8-9 us, 24 times per step)
  float sv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) sv[u] = 0.f;
  if (c < ncols) {
    int b = rl;
    for (; b + 240 < nblk; b += 256) {
#pragma unroll
      for (int u = 0; u < 16; ++u) sv[u] += partial[(int64_t)(b + 16 * u) * stride + c];
    }
    for (; b < nblk; b += 16) sv[0] += partial[(int64_t)b * stride + c];
  }
#pragma unroll
  for (int w = 8; w > 0; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) sv[u] += sv[u + w];
  red[rl][cl] = sv[0];
  __syncthreads();
  if (rl == 0 && c < ncols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    // up to three destinations for one partial row (the gate backward's dwc | dbias | dbc): columns >= c1 go to out1, >= c2 to out2
    float* dst = (out2 && c >= c2) ? out2 + (c - c2) : (out1 && c >= c1) ? out1 + (c - c1) : out + c;
    *dst = accumulate ? *dst + t : t;
  }
}
#define MERGE_GRID(ncols) dim3((unsigned)(((ncols) + 15) / 16))

// =====================================================================================
// gate score: s[n] = sum_j (a ka)(b kb) wc[j] + bc          one wave per row
// =====================================================================================
__global__ __launch_bounds__(256) void gate_score_kernel(const float* __restrict__ ab, const float* __restrict__ wc,
                                                         const float* __restrict__ bc, float p, const uint64_t* seed,
                                                         uint64_t stream_a, uint64_t stream_b, int64_t N, int64_t D,
                                                         float* __restrict__ s, const int64_t* __restrict__ rng_row) {
  const int lane = threadIdx.x & 63;
  const bool drop = seed && p > 0.f;
  uint64_t ka = 0, kb = 0;
  float inv = 1.f;
  if (drop) {
    const uint64_t sd = *seed;
    ka = rng_key(sd, stream_a); kb = rng_key(sd, stream_b); inv = hw_rcp(1.f - p);
  }
  // one wave per row, grid-stride over rows; 16-byte loads (D % 4 == 0 on this path), wc kept in registers across rows
  const int64_t D4 = D >> 2;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  if ((D & 3) == 0 && D4 == 96) {
    // D = 384 (the shipped width): a row is 96 float4 per branch = 1.5 per lane -> TWO rows per wave and iteration are exactly 3 float4 per
    // lane and branch, no idle lanes (one row per iteration left every second pass half empty: a quarter of the issue slots of a launch
    // that is bound by its hash arithmetic), two rows' loads in flight. Lane l, pass t: float4 index l + 64 t of the row pair.
    float4 w[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) w[t] = reinterpret_cast<const float4*>(wc)[(lane + 64 * t) % 96];
    const float bias = bc[0];
    for (int64_t n0 = 2 * wave0; n0 < N; n0 += 2 * nwaves) {
      float accA = 0.f, accB = 0.f;
      float4 av[3], bv[3];
      int64_t rowi[3];
      int qi[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int g = lane + 64 * t;
        const int64_t n = n0 + (g >= 96 ? 1 : 0);
        qi[t] = g >= 96 ? g - 96 : g;
        rowi[t] = n < N ? n : n0;                     // (an odd last row: the second half re-reads the first, weight 0)
        const float4* ra = reinterpret_cast<const float4*>(ab + rowi[t] * 2 * D);
        av[t] = ra[qi[t]];
        bv[t] = ra[D4 + qi[t]];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float4 a = av[t], b = bv[t];
        if (drop) {
          const uint64_t base = (uint64_t)((rng_row ? rng_row[rowi[t]] : rowi[t]) * D + qi[t] * 4);
          float fa[4], fb[4];
          rng_keep4(ka, base, p, inv, fa);
          rng_keep4(kb, base, p, inv, fb);
          a.x *= fa[0]; a.y *= fa[1]; a.z *= fa[2]; a.w *= fa[3];
          b.x *= fb[0]; b.y *= fb[1]; b.z *= fb[2]; b.w *= fb[3];
        }
        const float v = a.x * b.x * w[t].x + a.y * b.y * w[t].y + a.z * b.z * w[t].z + a.w * b.w * w[t].w;
        const bool second = lane + 64 * t >= 96;
        accA += second ? 0.f : v;
        accB += second ? v : 0.f;
      }
      accA = wave_sum(accA);
      accB = wave_sum(accB);
      if (lane == 0) {
        s[n0] = accA + bias;
        if (n0 + 1 < N) s[n0 + 1] = accB + bias;
      }
    }
    return;
  }
  if ((D & 3) == 0 && D4 <= 128) {
    float4 w[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t q = lane + 64 * t;
      w[t] = q < D4 ? reinterpret_cast<const float4*>(wc)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bias = bc[0];
    for (int64_t n = wave0; n < N; n += nwaves) {
      const float4* ra = reinterpret_cast<const float4*>(ab + n * 2 * D);
      const float4* rb = ra + D4;
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int64_t q = lane + 64 * t;
        if (q < D4) {
          float4 a = ra[q], b = rb[q];
          if (drop) {
            const uint64_t base = (uint64_t)((rng_row ? rng_row[n] : n) * D + q * 4);
            float fa[4], fb[4];
            rng_keep4(ka, base, p, inv, fa);
            rng_keep4(kb, base, p, inv, fb);
            a.x *= fa[0]; a.y *= fa[1]; a.z *= fa[2]; a.w *= fa[3];
            b.x *= fb[0]; b.y *= fb[1]; b.z *= fb[2]; b.w *= fb[3];
          }
          acc += a.x * b.x * w[t].x + a.y * b.y * w[t].y + a.z * b.z * w[t].z + a.w * b.w * w[t].w;
        }
      }
      acc = wave_sum(acc);
      if (lane == 0) s[n] = acc + bias;
    }
    return;
  }
  for (int64_t n = wave0; n < N; n += nwaves) {
    const float* row = ab + n * 2 * D;
    float acc = 0.f;
    for (int64_t j = lane; j < D; j += 64) {
      float a = row[j], b = row[D + j];
      if (drop) {
        const uint64_t idx = (uint64_t)((rng_row ? rng_row[n] : n) * D + j);
        a *= rng_keep(ka, idx, p, inv);
        b *= rng_keep(kb, idx, p, inv);
      }
      acc += a * b * wc[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) s[n] = acc + bc[0];
  }
}

extern "C" int advmil_gate_score_fwd(const float* ab, const float* wc, const float* bc, float drop_p, const uint64_t* seed,
                                     uint64_t stream_a, uint64_t stream_b, int64_t N, int64_t D, float* s,
                                     const int64_t* rng_row, advmil_stream_t stream) {
  if (!ab || !wc || !bc || !s || N <= 0 || D <= 0) return ADVMIL_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !seed)) return ADVMIL_EINVAL;
  int64_t blocks = (N + 3) / 4;
  if (blocks > 8192) blocks = 8192;          // grid-stride over rows: 32 workgroups per CU
  hipLaunchKernelGGL(gate_score_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ab, wc, bc,
                     drop_p, seed, stream_a, stream_b, N, D, s, rng_row);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// =====================================================================================
// softmax over instances + weighted pooling
// =====================================================================================
// Segments: a step batch is a ragged slab of bags; segment b owns rows [seg_ptr[b], seg_ptr[b+1]) (seg_ptr == NULL: one
// segment [0, N)). gridDim.y (or .x for the stats kernel) indexes the segment.
__device__ __forceinline__ void seg_range(const int64_t* seg_ptr, int b, int64_t N, int64_t& beg, int64_t& end) {
  beg = seg_ptr ? seg_ptr[b] : 0;
  end = seg_ptr ? seg_ptr[b + 1] : N;
}

// stats[2b] = max_n s, stats[2b+1] = 1 / sum_n exp(s - max) over segment b
__global__ __launch_bounds__(1024) void softmax_stats_kernel(const float* __restrict__ s, int64_t N,
                                                             const int64_t* __restrict__ seg_ptr, float* __restrict__ stats) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int64_t beg, end;
  seg_range(seg_ptr, blockIdx.x, N, beg, end);
  float mx = -INFINITY;
  for (int64_t n = beg + tid; n < end; n += 1024) mx = fmaxf(mx, s[n]);
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = red[0];
  for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
  __syncthreads();
  float sum = 0.f;
  for (int64_t n = beg + tid; n < end; n += 1024) sum += hw_exp(s[n] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[w] = sum;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k];
    stats[2 * blockIdx.x] = mx;
    stats[2 * blockIdx.x + 1] = hw_rcp(t);
  }
}

__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ s, const float* __restrict__ h,
                                                           int64_t ldh, int64_t N, int64_t D,
                                                           const int64_t* __restrict__ seg_ptr,
                                                           const float* __restrict__ stats, float* __restrict__ A,
                                                           float* __restrict__ partial, int rpb) {
  __shared__ __attribute__((aligned(16))) float red[1024];
  const RowColMap m = make_map(D);
  const int b = blockIdx.y;
  int64_t beg, end;
  seg_range(seg_ptr, b, N, beg, end);
  const float mx = stats[2 * b], inv = stats[2 * b + 1];
  const int64_t r0 = beg + (int64_t)blockIdx.x * rpb;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.active) {
    for (int r = m.r; r < rpb; r += m.rpp) {
      const int64_t n = r0 + r;
      if (n >= end) break;
      const float w = hw_exp(s[n] - mx) * inv;
      if (m.c4 == 0) A[n] = w;
      const float4 v = *reinterpret_cast<const float4*>(h + n * ldh + m.c4 * 4);
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
  }
  const float4 t = reduce_rows(acc, m, red);   // blocks past the segment's end contribute zeros
  if (m.active && m.r == 0)
    *reinterpret_cast<float4*>(partial + ((int64_t)b * gridDim.x + blockIdx.x) * D + m.c4 * 4) = t;
}

// Same contract as pool_partial_kernel for D % 8 == 0: every thread owns 8 consecutive columns (two 16-byte loads per row), so at
// D = 384 a workgroup walks 5 rows per pass with 240 of its 256 threads (the float4 mapping: 2 rows, 192 threads), and the row loop
// is unrolled with predicates so that 4 rows of loads are in flight per thread. Measured on the 16 x 8192 x 384 slab: the pooling
// call (statistics + this + merge) 61.9 us -> see DESIGN.md section 5 (pool_roofline of the bench line).
__global__ __launch_bounds__(256) void pool_partial8_kernel(const float* __restrict__ s, const float* __restrict__ h, int64_t ldh,
                                                            int64_t N, int64_t D, const int64_t* __restrict__ seg_ptr,
                                                            const float* __restrict__ stats, float* __restrict__ A,
                                                            float* __restrict__ partial, int rpb) {
  __shared__ __attribute__((aligned(16))) float red[2048];      // rpp * D <= 256 / (D/8) * D = 2048 floats
  const int cols8 = (int)(D >> 3), rpp = 256 / cols8;
  const int c8 = threadIdx.x % cols8, rl = threadIdx.x / cols8;
  const int b = blockIdx.y;
  int64_t beg, end;
  seg_range(seg_ptr, b, N, beg, end);
  const float mx = stats[2 * b], inv = stats[2 * b + 1];
  const int64_t r0 = beg + (int64_t)blockIdx.x * rpb;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (rl < rpp) {
#pragma unroll 4
    for (int r = rl; r < rpb; r += rpp) {
      const int64_t n = r0 + r;
      const bool ok = n < end;
      const int64_t nn = ok ? n : beg;                     // predicated: keeps the unrolled loads independent of the bound
      const float w = ok ? hw_exp(s[nn] - mx) * inv : 0.f;
      const float4 v0 = *reinterpret_cast<const float4*>(h + nn * ldh + c8 * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(h + nn * ldh + c8 * 8 + 4);
      if (ok && c8 == 0) A[n] = w;
      acc[0] += w * v0.x; acc[1] += w * v0.y; acc[2] += w * v0.z; acc[3] += w * v0.w;
      acc[4] += w * v1.x; acc[5] += w * v1.y; acc[6] += w * v1.z; acc[7] += w * v1.w;
    }
    float* dst = red + (int64_t)rl * D + c8 * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  __syncthreads();
  if ((int)threadIdx.x < (int)(D >> 2)) {                     // blocks past the segment's end contribute zeros
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < rpp; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(red + (int64_t)q * D + threadIdx.x * 4);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + ((int64_t)b * gridDim.x + blockIdx.x) * D + threadIdx.x * 4) = t;
  }
}

// ---- two-launch form of the pooling call (D % 8 == 0): no separate statistics pass.
// Launch 1: every workgroup softmax-weights ITS rows against its own maximum (online softmax: m_j = max of the block's scores,
// w_n = exp(s_n - m_j) kept in LDS -- one exponential per row instead of one per thread and row --, l_j = sum w_n) and leaves
// partial_j = sum w_n h_n with (m_j, l_j). Launch 2, per segment and 16-column block: M = max m_j, c_j = exp(m_j - M), L = sum l_j c_j
// (fixed order: deterministic, and identical in every workgroup of the segment), pooled = sum_j c_j partial_j / L, and the workgroups of
// a segment share out its rows to write A_n = exp(s_n - M) / L. Same contract as softmax_stats + pool_partial8 + colsum_merge.
// MEAN: a second partial row per workgroup, the UNWEIGHTED sum of its rows (the per-bag mean of h from the same pass over h: the
// projection discriminator's region-level inner product, GANSurv.py:96-98, needs mean_r(fc_ins) beside the pooled fc_ins)
// PL: h arrives as its bf16x3 operand planes (h = the hi plane's address, hlo = the lo plane's; x = hi + lo, exact in fp32): the pooled
// tensor of a slab whose producer wrote planes only (round 6: the generator's first layer leaves no fp32 copy -- 201 MB less written per
// launch and the rows just read by the gate contraction are still in the Infinity Cache). Same 4 bytes per element, same loads per row.
__device__ __forceinline__ void planes8(const bf16raw* __restrict__ hi, const bf16raw* __restrict__ lo, int64_t off, float4& v0, float4& v1) {
  const uint4 a = *reinterpret_cast<const uint4*>(hi + off);
  const uint4 b = *reinterpret_cast<const uint4*>(lo + off);
  v0.x = __uint_as_float(a.x << 16) + __uint_as_float(b.x << 16);
  v0.y = __uint_as_float(a.x & 0xffff0000u) + __uint_as_float(b.x & 0xffff0000u);
  v0.z = __uint_as_float(a.y << 16) + __uint_as_float(b.y << 16);
  v0.w = __uint_as_float(a.y & 0xffff0000u) + __uint_as_float(b.y & 0xffff0000u);
  v1.x = __uint_as_float(a.z << 16) + __uint_as_float(b.z << 16);
  v1.y = __uint_as_float(a.z & 0xffff0000u) + __uint_as_float(b.z & 0xffff0000u);
  v1.z = __uint_as_float(a.w << 16) + __uint_as_float(b.w << 16);
  v1.w = __uint_as_float(a.w & 0xffff0000u) + __uint_as_float(b.w & 0xffff0000u);
}

template <bool MEAN, bool PL = false>
__global__ __launch_bounds__(256) void pool_partial8_online_kernel(const float* __restrict__ s, const float* __restrict__ h, int64_t ldh,
                                                                   int64_t N, int64_t D, const int64_t* __restrict__ seg_ptr,
                                                                   float* __restrict__ partial, float* __restrict__ pstats, int rpb,
                                                                   float* __restrict__ mpartial, const bf16raw* __restrict__ hlo = nullptr) {
  __shared__ __attribute__((aligned(16))) float red[2048];      // rpp * D <= 256 / (D/8) * D = 2048 floats
  __shared__ float wts[512];                                    // rows_per_block <= 512
  __shared__ float wred[8];
  const int cols8 = (int)(D >> 3), rpp = 256 / cols8;
  const int c8 = threadIdx.x % cols8, rl = threadIdx.x / cols8;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.y;
  int64_t beg, end;
  seg_range(seg_ptr, b, N, beg, end);
  const int64_t r0 = beg + (int64_t)blockIdx.x * rpb;
  float* pst = pstats + ((int64_t)b * gridDim.x + blockIdx.x) * 2;
  float* prow = partial + ((int64_t)b * gridDim.x + blockIdx.x) * D;
  float* mrow = MEAN ? mpartial + ((int64_t)b * gridDim.x + blockIdx.x) * D : nullptr;
  if (r0 >= end) {                                              // past the segment's end: an empty block (weight 0 in the merge)
    if (tid < (int)(D >> 2)) {
      *reinterpret_cast<float4*>(prow + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MEAN) *reinterpret_cast<float4*>(mrow + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) { pst[0] = -INFINITY; pst[1] = 0.f; }
    return;
  }
  float mx = -INFINITY;
  for (int r = tid; r < rpb; r += 256) {
    const int64_t n = r0 + r;
    const float v = n < end ? s[n] : -INFINITY;
    wts[r] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if (lane == 0) wred[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  float ls = 0.f;
  for (int r = tid; r < rpb; r += 256) {
    const float e = (r0 + r < end) ? hw_exp(wts[r] - mx) : 0.f;
    wts[r] = e;
    ls += e;
  }
  ls = wave_sum(ls);
  if (lane == 0) wred[4 + w] = ls;
  __syncthreads();
  if (tid == 0) { pst[0] = mx; pst[1] = (wred[4] + wred[5]) + (wred[6] + wred[7]); }
  float acc[8], macc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j] = 0.f; macc[j] = 0.f; }
  if (rl < rpp) {
#pragma unroll 4
    for (int r = rl; r < rpb; r += rpp) {
      const int64_t n = r0 + r;
      const int64_t nn = n < end ? n : beg;                // predicated: keeps the unrolled loads independent of the bound (weight 0)
      const float wgt = wts[r];
      float4 v0, v1;
      if constexpr (PL) {
        planes8(reinterpret_cast<const bf16raw*>(h), hlo, nn * ldh + c8 * 8, v0, v1);
      } else {
        v0 = *reinterpret_cast<const float4*>(h + nn * ldh + c8 * 8);
        v1 = *reinterpret_cast<const float4*>(h + nn * ldh + c8 * 8 + 4);
      }
      acc[0] += wgt * v0.x; acc[1] += wgt * v0.y; acc[2] += wgt * v0.z; acc[3] += wgt * v0.w;
      acc[4] += wgt * v1.x; acc[5] += wgt * v1.y; acc[6] += wgt * v1.z; acc[7] += wgt * v1.w;
      if (MEAN) {
        const float one = n < end ? 1.f : 0.f;
        macc[0] += one * v0.x; macc[1] += one * v0.y; macc[2] += one * v0.z; macc[3] += one * v0.w;
        macc[4] += one * v1.x; macc[5] += one * v1.y; macc[6] += one * v1.z; macc[7] += one * v1.w;
      }
    }
    float* dst = red + (int64_t)rl * D + c8 * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  __syncthreads();
  if (tid < (int)(D >> 2)) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < rpp; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(red + (int64_t)q * D + tid * 4);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(prow + tid * 4) = t;
  }
  if (MEAN) {
    __syncthreads();
    if (rl < rpp) {
      float* dst = red + (int64_t)rl * D + c8 * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(macc[0], macc[1], macc[2], macc[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(macc[4], macc[5], macc[6], macc[7]);
    }
    __syncthreads();
    if (tid < (int)(D >> 2)) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int q = 0; q < rpp; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(red + (int64_t)q * D + tid * 4);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      *reinterpret_cast<float4*>(mrow + tid * 4) = t;
    }
  }
}

#define POOL_ONLINE_MAX_NBLK 2048
__global__ __launch_bounds__(256) void pool_merge_online_kernel(const float* __restrict__ partial, const float* __restrict__ pstats, int nblk,
                                                                int64_t D, const float* __restrict__ s, int64_t N,
                                                                const int64_t* __restrict__ seg_ptr, float* __restrict__ pooled,
                                                                float* __restrict__ A, float* __restrict__ stats,
                                                                const float* __restrict__ mpartial = nullptr, float* __restrict__ mean = nullptr) {
  __shared__ float sc[POOL_ONLINE_MAX_NBLK];
  __shared__ float red[16][17];
  __shared__ float wred[8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.y;
  const float* pst = pstats + (int64_t)b * nblk * 2;
  float mx = -INFINITY;
  for (int j = tid; j < nblk; j += 256) mx = fmaxf(mx, pst[2 * j]);
  mx = wave_max(mx);
  if (lane == 0) wred[w] = mx;
  __syncthreads();
  const float M = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  float ls = 0.f;
  for (int j = tid; j < nblk; j += 256) {
    const float mj = pst[2 * j];
    const float c = mj == -INFINITY ? 0.f : hw_exp(mj - M);
    sc[j] = c;
    ls += pst[2 * j + 1] * c;
  }
  ls = wave_sum(ls);
  if (lane == 0) wred[4 + w] = ls;
  __syncthreads();
  const float inv = hw_rcp((wred[4] + wred[5]) + (wred[6] + wred[7]));
  if (blockIdx.x == 0 && tid == 0 && stats) { stats[2 * b] = M; stats[2 * b + 1] = inv; }
  // pooled: this workgroup's 16 columns over the segment's partial rows (16 row-lanes, folded in order)
  const int cl = tid & 15, rl = tid >> 4;
  const int64_t c = (int64_t)blockIdx.x * 16 + cl;
  const float* pp = partial + (int64_t)b * nblk * D;
  float sv[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < D) {
    int j = rl;
    for (; j + 48 < nblk; j += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) sv[u] += pp[(int64_t)(j + 16 * u) * D + c] * sc[j + 16 * u];
    }
    for (; j < nblk; j += 16) sv[0] += pp[(int64_t)j * D + c] * sc[j];
  }
  red[rl][cl] = (sv[0] + sv[1]) + (sv[2] + sv[3]);
  __syncthreads();
  if (rl == 0 && c < D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    pooled[(int64_t)b * D + c] = t * inv;
  }
  int64_t beg, end;
  seg_range(seg_ptr, b, N, beg, end);
  if (mean) {        // the unweighted partial rows: mean = their sum / the segment's length (same column map, same fixed order)
    __syncthreads();
    const float* mp = mpartial + (int64_t)b * nblk * D;
    float mv[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      int j = rl;
      for (; j + 48 < nblk; j += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) mv[u] += mp[(int64_t)(j + 16 * u) * D + c];
      }
      for (; j < nblk; j += 16) mv[0] += mp[(int64_t)j * D + c];
    }
    red[rl][cl] = (mv[0] + mv[1]) + (mv[2] + mv[3]);
    __syncthreads();
    if (rl == 0 && c < D) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r][cl];
      mean[(int64_t)b * D + c] = end > beg ? t / (float)(end - beg) : 0.f;
    }
  }
  // A: the workgroups of the segment share out its rows
  for (int64_t n = beg + (int64_t)blockIdx.x * 256 + tid; n < end; n += (int64_t)gridDim.x * 256) A[n] = hw_exp(s[n] - M) * inv;
}

static inline int64_t pool_nblk(int64_t max_len) { const int r = rows_per_block(max_len); return (max_len + r - 1) / r; }

extern "C" size_t advmil_softmax_pool_workspace_bytes(int64_t max_len, int64_t D, int nseg) {
  if (nseg < 1) nseg = 1;
  const int64_t a = 4 * (int64_t)nseg + (int64_t)nseg * pool_nblk(max_len) * (D + 2);   // fwd: stats + partials (+ per-block (m, l))
  const int64_t b = 4 + (int64_t)nseg * ((max_len + 3) / 4);                       // bwd: per-workgroup partials of sum A t
  return (size_t)(a > b ? a : b) * sizeof(float);
}

static int softmax_pool_fwd_impl(const float* s, const float* h, int64_t ldh, int64_t N, int64_t D, int nseg, const int64_t* seg_ptr,
                                 int64_t max_len, float* A, float* pooled, float* mean, void* ws, size_t ws_bytes, hipStream_t stream,
                                 const bf16raw* hlo = nullptr);

extern "C" int advmil_softmax_pool_fwd(const float* s, const float* h, int64_t ldh, int64_t N, int64_t D, int nseg,
                                       const int64_t* seg_ptr, int64_t max_len, float* A, float* pooled, void* ws,
                                       size_t ws_bytes, advmil_stream_t stream_) {
  return softmax_pool_fwd_impl(s, h, ldh, N, D, nseg, seg_ptr, max_len, A, pooled, nullptr, ws, ws_bytes, (hipStream_t)stream_);
}

extern "C" int advmil_softmax_pool_fwd_planes(const float* s, const void* h_hi, const void* h_lo, int64_t ldh, int64_t N, int64_t D, int nseg,
                                              const int64_t* seg_ptr, int64_t max_len, float* A, float* pooled, void* ws,
                                              size_t ws_bytes, advmil_stream_t stream_) {
  if (!h_hi || !h_lo) return ADVMIL_EINVAL;
  return softmax_pool_fwd_impl(s, (const float*)h_hi, ldh, N, D, nseg, seg_ptr, max_len, A, pooled, nullptr, ws, ws_bytes, (hipStream_t)stream_,
                               (const bf16raw*)h_lo);
}

// the unweighted partial rows sit behind the plain call's workspace, on a 16-byte boundary (they are stored as float4)
static inline size_t pool_mean_offset_bytes(int64_t max_len, int64_t D, int nseg) {
  return (advmil_softmax_pool_workspace_bytes(max_len, D, nseg) + 15) / 16 * 16;
}

extern "C" size_t advmil_softmax_pool_mean_workspace_bytes(int64_t max_len, int64_t D, int nseg) {
  if (nseg < 1) nseg = 1;
  return pool_mean_offset_bytes(max_len, D, nseg) + (size_t)((int64_t)nseg * pool_nblk(max_len) * D) * sizeof(float);
}

extern "C" int advmil_softmax_pool_mean_fwd(const float* s, const float* h, int64_t ldh, int64_t N, int64_t D, int nseg,
                                            const int64_t* seg_ptr, int64_t max_len, float* A, float* pooled, float* mean, void* ws,
                                            size_t ws_bytes, advmil_stream_t stream_) {
  if (!mean) return ADVMIL_EINVAL;
  return softmax_pool_fwd_impl(s, h, ldh, N, D, nseg, seg_ptr, max_len, A, pooled, mean, ws, ws_bytes, (hipStream_t)stream_);
}

static int softmax_pool_fwd_impl(const float* s, const float* h, int64_t ldh, int64_t N, int64_t D, int nseg, const int64_t* seg_ptr,
                                 int64_t max_len, float* A, float* pooled, float* mean, void* ws, size_t ws_bytes, hipStream_t stream,
                                 const bf16raw* hlo) {
  if (!s || !h || !A || !pooled || !ws || N <= 0 || D <= 0 || (D & 3) || D > 1024 || (ldh & 3)) return ADVMIL_EINVAL;
  if (ldh < D || ((uintptr_t)h & 15)) return ADVMIL_EINVAL;        // rows are read as 16-byte vectors
  // planes: 8 halfwords per 16-byte load -> D and the pitch multiples of 8; the two-launch form only; no unweighted mean beside it
  if (hlo && (((uintptr_t)hlo & 15) || (D & 7) || (ldh & 7) || D < 16 || mean)) return ADVMIL_EINVAL;
  if (!seg_ptr && nseg > 1) return ADVMIL_EINVAL;                   // several bags need their row offsets
  if (!seg_ptr) { nseg = 1; max_len = N; }
  if (nseg < 1 || max_len <= 0 || max_len > N) return ADVMIL_EINVAL;
  if (ws_bytes < (mean ? advmil_softmax_pool_mean_workspace_bytes(max_len, D, nseg) : advmil_softmax_pool_workspace_bytes(max_len, D, nseg)))
    return ADVMIL_EWORKSPACE;
  float* stats = (float*)ws;
  float* partial = stats + 4 * nseg;
  const int nblk = (int)pool_nblk(max_len);
  if ((D & 7) == 0 && D >= 16 && nblk <= POOL_ONLINE_MAX_NBLK) {     // two launches: online-softmax partials, then merge + A
    float* pstats = partial + (int64_t)nseg * nblk * D;
    if (mean) {
      float* mpartial = (float*)ws + pool_mean_offset_bytes(max_len, D, nseg) / sizeof(float);
      hipLaunchKernelGGL((pool_partial8_online_kernel<true, false>), dim3(nblk, nseg), dim3(256), 0, stream, s, h, ldh, N, D, seg_ptr, partial, pstats,
                         rows_per_block(max_len), mpartial, (const bf16raw*)nullptr);
      ADVMIL_LAUNCH_CHECK();
      hipLaunchKernelGGL(pool_merge_online_kernel, dim3((unsigned)((D + 15) / 16), nseg), dim3(256), 0, stream, partial, pstats, nblk, D, s,
                         N, seg_ptr, pooled, A, stats, (const float*)mpartial, mean);
      ADVMIL_LAUNCH_CHECK();
      return ADVMIL_OK;
    }
    if (hlo)
      hipLaunchKernelGGL((pool_partial8_online_kernel<false, true>), dim3(nblk, nseg), dim3(256), 0, stream, s, h, ldh, N, D, seg_ptr, partial,
                         pstats, rows_per_block(max_len), (float*)nullptr, hlo);
    else
      hipLaunchKernelGGL((pool_partial8_online_kernel<false, false>), dim3(nblk, nseg), dim3(256), 0, stream, s, h, ldh, N, D, seg_ptr, partial,
                         pstats, rows_per_block(max_len), (float*)nullptr, (const bf16raw*)nullptr);
    ADVMIL_LAUNCH_CHECK();
    hipLaunchKernelGGL(pool_merge_online_kernel, dim3((unsigned)((D + 15) / 16), nseg), dim3(256), 0, stream, partial, pstats, nblk, D, s,
                       N, seg_ptr, pooled, A, stats, (const float*)nullptr, (float*)nullptr);
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  if (mean || hlo) return ADVMIL_EINVAL;   // (the mean / the plane-held h ride in the two-launch form only: D % 8 == 0)
  hipLaunchKernelGGL(softmax_stats_kernel, dim3(nseg), dim3(1024), 0, stream, s, N, seg_ptr, stats);
  ADVMIL_LAUNCH_CHECK();
  if ((D & 7) == 0 && D >= 16)
    hipLaunchKernelGGL(pool_partial8_kernel, dim3(nblk, nseg), dim3(256), 0, stream, s, h, ldh, N, D, seg_ptr, stats, A, partial,
                       rows_per_block(max_len));
  else
    hipLaunchKernelGGL(pool_partial_kernel, dim3(nblk, nseg), dim3(256), 0, stream, s, h, ldh, N, D, seg_ptr, stats, A, partial,
                       rows_per_block(max_len));
  ADVMIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_merge_kernel, dim3((unsigned)((D + 15) / 16), nseg), dim3(256), 0, stream, partial, nblk, D, D, pooled,
                     0, (int64_t)nblk * D, D);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// backward stage 1: t[n] = dA[n] + dot(dpooled[seg], h[n]); partial[seg][wg] = sum over the wg's 16 rows of A[n] t[n].
// One wave per FOUR rows, their loads issued together (one row per wave left a single 16-byte load in flight per lane: 0.52 of the
// HBM roof at the 16-bag slab).
#define PBD_ROWS 4
template <bool PL>
__global__ __launch_bounds__(256) void pool_bwd_dot_kernel(const float* __restrict__ dp, const float* __restrict__ dA,
                                                           const float* __restrict__ A, const float* __restrict__ h,
                                                           int64_t ldh, int64_t N, int64_t D,
                                                           const int64_t* __restrict__ seg_ptr, float* __restrict__ t,
                                                           float* __restrict__ partial, const bf16raw* __restrict__ hlo) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y;
  int64_t beg, end;
  seg_range(seg_ptr, b, N, beg, end);
  const int64_t n0 = beg + ((int64_t)blockIdx.x * 4 + w) * PBD_ROWS;
  const float* dpb = dp + (int64_t)b * D;
  float acc[PBD_ROWS];
#pragma unroll
  for (int rr = 0; rr < PBD_ROWS; ++rr) acc[rr] = 0.f;
  if (PL && n0 < end) {
    // h as planes: 16 bytes of each plane = 8 columns per load. The wave's 4 rows x D / 8 chunks are dealt out lane by lane (D = 384:
    // 192 chunks = 3 per lane, no idle lanes; 8-byte loads of 4 columns measured 53 us against the fp32 kernel's 42 at the 16-bag slab)
    const int nch = (int)(D >> 3), items = PBD_ROWS * nch;
    for (int g = lane; g < items; g += 64) {
      const int rr = g / nch, ch = g - rr * nch;
      if (n0 + rr < end) {
        float4 v0, v1;
        planes8(reinterpret_cast<const bf16raw*>(h), hlo, (n0 + rr) * ldh + ch * 8, v0, v1);
        const float4 a0 = *reinterpret_cast<const float4*>(dpb + ch * 8), a1 = *reinterpret_cast<const float4*>(dpb + ch * 8 + 4);
        const float d = a0.x * v0.x + a0.y * v0.y + a0.z * v0.z + a0.w * v0.w + a1.x * v1.x + a1.y * v1.y + a1.z * v1.z + a1.w * v1.w;
#pragma unroll
        for (int k = 0; k < PBD_ROWS; ++k) acc[k] += (k == rr) ? d : 0.f;
      }
    }
  } else if (n0 < end) {
    if ((D & 3) == 0 && (ldh & 3) == 0) {            // 16-byte loads (D = 384 -> 96 float4 per row, 1.5 per lane)
      const float4* d4 = reinterpret_cast<const float4*>(dpb);
      for (int64_t q = lane; q < (D >> 2); q += 64) {
        const float4 a = d4[q];
        float4 v[PBD_ROWS];
#pragma unroll
        for (int rr = 0; rr < PBD_ROWS; ++rr)
          v[rr] = (n0 + rr < end) ? reinterpret_cast<const float4*>(h + (n0 + rr) * ldh)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int rr = 0; rr < PBD_ROWS; ++rr) acc[rr] += a.x * v[rr].x + a.y * v[rr].y + a.z * v[rr].z + a.w * v[rr].w;
      }
    } else {
      for (int64_t j = lane; j < D; j += 64) {
        const float a = dpb[j];
#pragma unroll
        for (int rr = 0; rr < PBD_ROWS; ++rr)
          if (n0 + rr < end) acc[rr] += a * h[(n0 + rr) * ldh + j];
      }
    }
  }
  float contrib = 0.f;
#pragma unroll
  for (int rr = 0; rr < PBD_ROWS; ++rr) {
    const int64_t n = n0 + rr;
    float s = wave_sum(acc[rr]);
    if (n < end) {
      if (dA) s += dA[n];
      if (lane == 0) t[n] = s;
      contrib += A[n] * s;
    }
  }
  if (lane == 0) red[w] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// backward stage 2: c = sum partial[seg]; ds[n] = A[n] (t[n] - c)   (t aliases ds)
__global__ __launch_bounds__(256) void pool_bwd_ds_kernel(const float* __restrict__ A, const float* __restrict__ partial,
                                                          int npart, int64_t N, const int64_t* __restrict__ seg_ptr,
                                                          float* __restrict__ ds) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  int64_t beg, end;
  seg_range(seg_ptr, b, N, beg, end);
  float c = 0.f;
  for (int k = threadIdx.x; k < npart; k += 256) c += partial[(int64_t)b * npart + k];
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  c = red[0] + red[1] + red[2] + red[3];
  const int64_t n = beg + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n < end) ds[n] = A[n] * (ds[n] - c);
}

static int softmax_pool_bwd_impl(const float* dpooled, const float* dA, const float* A, const float* h, const bf16raw* hlo, int64_t ldh,
                                 int64_t N, int64_t D, int nseg, const int64_t* seg_ptr, int64_t max_len, float* ds,
                                 void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!dpooled || !A || !h || !ds || !ws || N <= 0 || D <= 0) return ADVMIL_EINVAL;
  if (ldh < D || (ldh & 3) || ((uintptr_t)h & 15)) return ADVMIL_EINVAL;
  if (hlo && (((uintptr_t)hlo & 15) || (D & 7) || (ldh & 7) || (((uintptr_t)dpooled) & 15))) return ADVMIL_EINVAL;
  if (!seg_ptr && nseg > 1) return ADVMIL_EINVAL;
  if (!seg_ptr) { nseg = 1; max_len = N; }
  if (nseg < 1 || max_len <= 0 || max_len > N) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_softmax_pool_workspace_bytes(max_len, D, nseg)) return ADVMIL_EWORKSPACE;
  float* partial = (float*)ws + 4;
  const int nwg = (int)((max_len + 4 * PBD_ROWS - 1) / (4 * PBD_ROWS));
  if (hlo) hipLaunchKernelGGL(pool_bwd_dot_kernel<true>, dim3(nwg, nseg), dim3(256), 0, stream, dpooled, dA, A, h, ldh, N, D, seg_ptr, ds, partial, hlo);
  else hipLaunchKernelGGL(pool_bwd_dot_kernel<false>, dim3(nwg, nseg), dim3(256), 0, stream, dpooled, dA, A, h, ldh, N, D, seg_ptr, ds, partial, hlo);
  ADVMIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(pool_bwd_ds_kernel, dim3((unsigned)((max_len + 255) / 256), nseg), dim3(256), 0, stream, A, partial, nwg, N,
                     seg_ptr, ds);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_softmax_pool_bwd(const float* dpooled, const float* dA, const float* A, const float* h, int64_t ldh,
                                       int64_t N, int64_t D, int nseg, const int64_t* seg_ptr, int64_t max_len, float* ds,
                                       void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  return softmax_pool_bwd_impl(dpooled, dA, A, h, nullptr, ldh, N, D, nseg, seg_ptr, max_len, ds, ws, ws_bytes, (hipStream_t)stream_);
}
extern "C" int advmil_softmax_pool_bwd_planes(const float* dpooled, const float* dA, const float* A, const void* h_hi, const void* h_lo,
                                              int64_t ldh, int64_t N, int64_t D, int nseg, const int64_t* seg_ptr, int64_t max_len,
                                              float* ds, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  if (!h_hi || !h_lo) return ADVMIL_EINVAL;
  return softmax_pool_bwd_impl(dpooled, dA, A, (const float*)h_hi, (const bf16raw*)h_lo, ldh, N, D, nseg, seg_ptr, max_len, ds, ws, ws_bytes,
                               (hipStream_t)stream_);
}

// =====================================================================================
// Train-mode dropout of a tensor held as operand planes: out = split(dropout(in_hi + in_lo)) as planes again, plus one bit per element
// (out > 0). The replay of the generator's memoized first layer (ops.ForwardMemo) when that layer left planes only: 8 bytes per element
// moved instead of 12 (fp32 in, fp32 + planes out), same draw as advmil_act_dropout_bwd's replay (stream, element index m * N + n).
// One thread = 8 consecutive columns: 16 bytes of each plane in, 16 bytes of each plane and one BYTE of the bit words out.
// =====================================================================================
__global__ __launch_bounds__(256) void dropout_planes_kernel(const bf16raw* __restrict__ ihi, const bf16raw* __restrict__ ilo, int64_t M, int64_t N,
                                                             float p, const uint64_t* __restrict__ seed, uint64_t stream_id,
                                                             const int64_t* __restrict__ rng_row, bf16raw* __restrict__ ohi,
                                                             bf16raw* __restrict__ olo, uint8_t* __restrict__ bits, float pg,
                                                             uint64_t stream_ga, uint64_t stream_gb, uint8_t* __restrict__ gbits_a,
                                                             uint8_t* __restrict__ gbits_b) {
  const uint64_t key = rng_key(*seed, stream_id);
  const float inv = hw_rcp(1.f - p);
  // (optional) the keep bits of the gated attention scorer's two branch dropouts over the SAME [M, N] index space, drawn here -- a launch
  // bound by its memory traffic -- so that the gate contraction's epilogue, where the matrix pipe would idle, only tests bits
  const uint64_t kga = gbits_a ? rng_key(*seed, stream_ga) : 0, kgb = gbits_a ? rng_key(*seed, stream_gb) : 0;
  const int64_t n8 = N >> 3, total = M * n8;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / n8, c = (e % n8) * 8;
    float4 v0, v1;
    planes8(ihi, ilo, row * N + c, v0, v1);
    const uint64_t base = (uint64_t)((rng_row ? rng_row[row] : row) * N + c);
    float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint32_t b = 0u;
    float f0[4], f1[4];
    rng_keep4(key, base, p, inv, f0);
    rng_keep4(key, base + 4, p, inv, f1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      x[q] *= q < 4 ? f0[q & 3] : f1[q & 3];
      b |= (x[q] > 0.f) ? (1u << q) : 0u;
    }
    uint2 h0, l0, h1, l1;
    split4(make_float4(x[0], x[1], x[2], x[3]), h0, l0);
    split4(make_float4(x[4], x[5], x[6], x[7]), h1, l1);
    *reinterpret_cast<uint4*>(ohi + row * N + c) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4*>(olo + row * N + c) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    if (bits) bits[row * (N >> 3) + (c >> 3)] = (uint8_t)b;       // little-endian bytes of the [M, N / 32] uint32 words
    if (gbits_a) {
      float fa0[4], fa1[4], fb0[4], fb1[4];
      rng_keep4(kga, base, pg, 1.f, fa0); rng_keep4(kga, base + 4, pg, 1.f, fa1);
      rng_keep4(kgb, base, pg, 1.f, fb0); rng_keep4(kgb, base + 4, pg, 1.f, fb1);
      uint32_t wa = 0u, wb = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        wa |= (fa0[q] != 0.f ? (1u << q) : 0u) | (fa1[q] != 0.f ? (16u << q) : 0u);
        wb |= (fb0[q] != 0.f ? (1u << q) : 0u) | (fb1[q] != 0.f ? (16u << q) : 0u);
      }
      gbits_a[row * (N >> 3) + (c >> 3)] = (uint8_t)wa;
      gbits_b[row * (N >> 3) + (c >> 3)] = (uint8_t)wb;
    }
  }
}

extern "C" int advmil_dropout_planes(const void* in_hi, const void* in_lo, int64_t M, int64_t N, float drop_p, const uint64_t* seed,
                                     uint64_t stream_id, const int64_t* rng_row, void* out_hi, void* out_lo, void* bits, float gate_p,
                                     uint64_t gate_stream_a, uint64_t gate_stream_b, void* gate_bits_a, void* gate_bits_b,
                                     advmil_stream_t stream_) {
  if (!in_hi || !in_lo || !out_hi || !out_lo || !seed || M <= 0 || N <= 0 || (N & 31) || !(drop_p > 0.f) || drop_p >= 1.f) return ADVMIL_EINVAL;
  if ((gate_bits_a != nullptr) != (gate_bits_b != nullptr) || (gate_bits_a && (!(gate_p > 0.f) || gate_p >= 1.f)) ||
      (((uintptr_t)gate_bits_a | (uintptr_t)gate_bits_b) & 3))
    return ADVMIL_EINVAL;
  if ((((uintptr_t)in_hi | (uintptr_t)in_lo | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) || ((uintptr_t)bits & 3)) return ADVMIL_EINVAL;
  int64_t blocks = (M * (N >> 3) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16raw*)in_hi, (const bf16raw*)in_lo,
                     M, N, drop_p, seed, stream_id, rng_row, (bf16raw*)out_hi, (bf16raw*)out_lo, (uint8_t*)bits, gate_p, gate_stream_a,
                     gate_stream_b, (uint8_t*)gate_bits_a, (uint8_t*)gate_bits_b);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// backward of the per-segment row mean / of any pooled = sum_n A[n] h[n] with constant A:  dh[n, :] = A[n] * dpooled[seg(n), :]
// (the region-level inner product of the projection discriminator, GANSurv.py:96-98: mean_r fc_ins_r; was index_select + mul)
__global__ __launch_bounds__(256) void seg_scale_rows_kernel(const float* __restrict__ dpooled, const float* __restrict__ A,
                                                             const int32_t* __restrict__ rowseg, int64_t N, int64_t D,
                                                             float* __restrict__ dh) {
  const int64_t q4 = D >> 2, total = N * q4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t n = e / q4, c = (e % q4) * 4;
    const float a = A[n];
    const float4 v = *reinterpret_cast<const float4*>(dpooled + (int64_t)(rowseg ? rowseg[n] : 0) * D + c);
    *reinterpret_cast<float4*>(dh + n * D + c) = make_float4(a * v.x, a * v.y, a * v.z, a * v.w);
  }
}
extern "C" int advmil_seg_scale_rows(const float* dpooled, const float* A, const int32_t* rowseg, int64_t N, int64_t D, float* dh,
                                     advmil_stream_t stream_) {
  if (!dpooled || !A || !dh || N <= 0 || D <= 0 || (D & 3) || (((uintptr_t)dpooled | (uintptr_t)dh) & 15)) return ADVMIL_EINVAL;
  int64_t blocks = (N * (D >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(seg_scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, dpooled, A, rowseg, N, D, dh);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// =====================================================================================
// Glue of the fused gate score (the contraction that reduces sum_j wc_j tanh(a_j) sigmoid(b_j) in its epilogue):
//   gate_interleave: rows a0, b0, a1, b1, ... of the two branch weights as one [2D, D] matrix (fp32 + its bf16x3 planes) and the
//     interleaved bias, so that a branch pair lands in adjacent accumulator columns -- one launch instead of stack, stack, split;
//   gate_partial_sum: s[n] = sum_j partial[n][j] + bc -- the per-column-block partials of that epilogue, one launch instead of sum + add.
// =====================================================================================
__global__ __launch_bounds__(256) void gate_interleave_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb,
                                                              const float* __restrict__ ba, const float* __restrict__ bb, int D,
                                                              float* __restrict__ Wi, bf16raw* __restrict__ hi, bf16raw* __restrict__ lo,
                                                              float* __restrict__ bi, int pair32) {
  const int q4 = D >> 2;                                    // float4 per row
  const int64_t total = (int64_t)2 * D * q4;
  // output row r <- branch `br`, unit j: element-interleaved (r = 2 j + br: the no-grad fused gate score) or, pair32, in blocks of 32
  // (r = 64 (j / 32) + 32 br + j % 32: the training form, whose stored activations the backward reads back block-wise)
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / q4), c = (int)(e % q4) * 4;
    const int br = pair32 ? ((r >> 5) & 1) : (r & 1), j = pair32 ? (((r >> 6) << 5) | (r & 31)) : (r >> 1);
    const float* src = br ? Wb : Wa;
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)j * D + c);
    *reinterpret_cast<float4*>(Wi + (int64_t)r * D + c) = v;
    if (hi) {
      uint2 hh, ll;
      split4(v, hh, ll);
      *reinterpret_cast<uint2*>(hi + (int64_t)r * D + c) = hh;
      *reinterpret_cast<uint2*>(lo + (int64_t)r * D + c) = ll;
    }
  }
  if (blockIdx.x == 0)
    for (int r = threadIdx.x; r < 2 * D; r += 256) {
      const int br = pair32 ? ((r >> 5) & 1) : (r & 1), j = pair32 ? (((r >> 6) << 5) | (r & 31)) : (r >> 1);
      bi[r] = br ? bb[j] : ba[j];
    }
}
__global__ __launch_bounds__(256) void gate_partial_sum_kernel(const float* __restrict__ partial, int np, const float* __restrict__ bc,
                                                               int64_t N, float* __restrict__ s) {
  const float b = bc ? bc[0] : 0.f;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    float a = 0.f;
    for (int j = 0; j < np; ++j) a += partial[n * np + j];     // column blocks in index order: same sum as torch.sum(dim=1) is NOT implied, see tests
    s[n] = a + b;
  }
}
extern "C" int advmil_gate_interleave(const float* Wa, const float* Wb, const float* ba, const float* bb, int D, float* Wi, void* Wi_hi,
                                      void* Wi_lo, float* bi, int pair32, advmil_stream_t stream_) {
  if (!Wa || !Wb || !ba || !bb || !Wi || !bi || D <= 0 || (D & 3) || ((Wi_hi != nullptr) != (Wi_lo != nullptr))) return ADVMIL_EINVAL;
  if (pair32 && (D & 31)) return ADVMIL_EINVAL;
  if (((uintptr_t)Wa & 15) || ((uintptr_t)Wb & 15) || ((uintptr_t)Wi & 15) || ((uintptr_t)Wi_hi & 7) || ((uintptr_t)Wi_lo & 7)) return ADVMIL_EINVAL;
  int64_t blocks = ((int64_t)2 * D * (D >> 2) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gate_interleave_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, Wa, Wb, ba, bb, D, Wi,
                     (bf16raw*)Wi_hi, (bf16raw*)Wi_lo, bi, pair32);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
extern "C" int advmil_gate_partial_sum(const float* partial, int np, const float* bc, int64_t N, float* s, advmil_stream_t stream_) {
  if (!partial || !s || np <= 0 || N <= 0) return ADVMIL_EINVAL;
  int64_t blocks = (N + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gate_partial_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, partial, np, bc, N, s);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// =====================================================================================
// gate backward: ds[N] -> dG[N,2D] (grads wrt the two pre-activations), dwc, dbc, dbias
// =====================================================================================
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ ab, const float* __restrict__ ds,
                                                       const float* __restrict__ wc, float p, const uint64_t* seed,
                                                       uint64_t stream_a, uint64_t stream_b, int64_t N, int64_t D,
                                                       float* __restrict__ dG, float* __restrict__ partial, int rpb,
                                                       const int64_t* __restrict__ rng_row, bf16raw* __restrict__ g_hi,
                                                       bf16raw* __restrict__ g_lo, int pair32) {
  __shared__ __attribute__((aligned(16))) float red[1024];
  const RowColMap m = make_map(D);
  // where unit j = 4 c4 .. of the two branches sits in a row of ab / dG: [a | b] halves, or (pair32, the fused training gate score's
  // layout) blocks of 32: a_j at 64 (j / 32) + j % 32, b_j 32 further
  const int64_t oa = pair32 ? (int64_t)(((m.c4 * 4) >> 5) << 6) + ((m.c4 * 4) & 31) : (int64_t)m.c4 * 4;
  const int64_t ob = pair32 ? oa + 32 : D + (int64_t)m.c4 * 4;
  const bool drop = seed && p > 0.f;
  uint64_t ka = 0, kb = 0;
  float inv = 1.f;
  if (drop) {
    const uint64_t sd = *seed;
    ka = rng_key(sd, stream_a); kb = rng_key(sd, stream_b); inv = hw_rcp(1.f - p);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  float4 s_wc = make_float4(0.f, 0.f, 0.f, 0.f), s_a = s_wc, s_b = s_wc;
  float s_ds = 0.f;
  if (m.active) {
    const float4 w4 = *reinterpret_cast<const float4*>(wc + m.c4 * 4);
    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
    for (int r = m.r; r < rpb; r += m.rpp) {
      const int64_t n = r0 + r;
      if (n >= N) break;
      const float d = ds[n];
      if (m.c4 == 0) s_ds += d;
      const float4 a4 = *reinterpret_cast<const float4*>(ab + n * 2 * D + oa);
      const float4 b4 = *reinterpret_cast<const float4*>(ab + n * 2 * D + ob);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
      float ga[4], gb[4], gw[4];
      float fa4[4] = {1.f, 1.f, 1.f, 1.f}, fb4[4] = {1.f, 1.f, 1.f, 1.f};
      if (drop) {
        const uint64_t idx = (uint64_t)((rng_row ? rng_row[n] : n) * D + m.c4 * 4);
        rng_keep4(ka, idx, p, inv, fa4);
        rng_keep4(kb, idx, p, inv, fb4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float fa = fa4[q], fb = fb4[q];
        const float ad = av[q] * fa, bd = bv[q] * fb;      // post-dropout branch values
        gw[q] = d * ad * bd;                               // d wc[j]
        ga[q] = d * wv[q] * bd * fa * (1.f - av[q] * av[q]);   // d pre_a  (tanh')
        gb[q] = d * wv[q] * ad * fb * bv[q] * (1.f - bv[q]);   // d pre_b  (sigmoid')
      }
      if (dG) {    // (NULL: the caller keeps dG as planes only -- its two consumers are bf16x3 contractions that would split it anyway)
        *reinterpret_cast<float4*>(dG + n * 2 * D + oa) = make_float4(ga[0], ga[1], ga[2], ga[3]);
        *reinterpret_cast<float4*>(dG + n * 2 * D + ob) = make_float4(gb[0], gb[1], gb[2], gb[3]);
      }
      if (g_hi) {   // bf16x3 operand planes of dG for the contractions that read it (dh = dG Wab, dWab = dG^T h)
        uint2 hh, ll;
        split4(make_float4(ga[0], ga[1], ga[2], ga[3]), hh, ll);
        *reinterpret_cast<uint2*>(g_hi + n * 2 * D + oa) = hh;
        *reinterpret_cast<uint2*>(g_lo + n * 2 * D + oa) = ll;
        split4(make_float4(gb[0], gb[1], gb[2], gb[3]), hh, ll);
        *reinterpret_cast<uint2*>(g_hi + n * 2 * D + ob) = hh;
        *reinterpret_cast<uint2*>(g_lo + n * 2 * D + ob) = ll;
      }
      s_wc.x += gw[0]; s_wc.y += gw[1]; s_wc.z += gw[2]; s_wc.w += gw[3];
      s_a.x += ga[0]; s_a.y += ga[1]; s_a.z += ga[2]; s_a.w += ga[3];
      s_b.x += gb[0]; s_b.y += gb[1]; s_b.z += gb[2]; s_b.w += gb[3];
    }
  }
  float* prow = partial + (int64_t)blockIdx.x * (3 * D + 4);
  float4 t = reduce_rows(s_wc, m, red);
  if (m.active && m.r == 0) *reinterpret_cast<float4*>(prow + m.c4 * 4) = t;
  t = reduce_rows(s_a, m, red);
  if (m.active && m.r == 0) *reinterpret_cast<float4*>(prow + D + m.c4 * 4) = t;
  t = reduce_rows(s_b, m, red);
  if (m.active && m.r == 0) *reinterpret_cast<float4*>(prow + 2 * D + m.c4 * 4) = t;
  // sum of ds over the block's rows (threads with c4 == 0 hold pieces)
  __syncthreads();
  if (m.active && m.c4 == 0) red[m.r] = s_ds;
  __syncthreads();
  if (threadIdx.x == 0) {
    float q = 0.f;
    for (int r = 0; r < m.rpp; ++r) q += red[r];
    prow[3 * D] = q;
  }
}

extern "C" size_t advmil_gate_bwd_workspace_bytes(int64_t N, int64_t D) {
  const int64_t nblk = (N + rows_per_block(N) - 1) / rows_per_block(N);
  return (size_t)(nblk * (3 * D + 4)) * sizeof(float);
}

extern "C" int advmil_gate_bwd(const float* ab, const float* ds, const float* wc, float drop_p, const uint64_t* seed,
                               uint64_t stream_a, uint64_t stream_b, int64_t N, int64_t D, float* dG, float* dwc,
                               float* dbc, float* dbias, int accumulate, const int64_t* rng_row, void* dG_hi, void* dG_lo, int pair32,
                               void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ab || !ds || !wc || (!dG && !dG_hi) || !dwc || !dbc || !dbias || !ws || N <= 0 || D <= 0 || (D & 3) || D > 1024 ||
      ((dG_hi != nullptr) != (dG_lo != nullptr)) || (pair32 && (D & 31)))
    return ADVMIL_EINVAL;
  if (ws_bytes < advmil_gate_bwd_workspace_bytes(N, D)) return ADVMIL_EWORKSPACE;
  const int rpb = rows_per_block(N);
  const int nblk = (int)((N + rpb - 1) / rpb);
  float* partial = (float*)ws;
  const int64_t stride = 3 * D + 4;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(nblk), dim3(256), 0, stream, ab, ds, wc, drop_p, seed, stream_a, stream_b, N, D,
                     dG, partial, rpb, rng_row, (bf16raw*)dG_hi, (bf16raw*)dG_lo, pair32);
  ADVMIL_LAUNCH_CHECK();
  // dwc | dbias(a) | dbias(b) | dbc are adjacent in the partial rows: one merge launch over the 3D+1 columns, three destinations
  { const int rc = advmil_sumq(stream, partial, nblk, stride, 3 * D + 1, dwc, accumulate, dbias, D, dbc, 3 * D); if (rc) return rc; }
  return ADVMIL_OK;
}

// =====================================================================================
// backward of y = dropout(act(pre)):  dpre = dy * keep * act'(y_pre_dropout);  dbias = colsum(dpre)
// =====================================================================================
__global__ __launch_bounds__(256) void act_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              int act, float p, const uint64_t* seed, uint64_t stream_id,
                                                              int64_t M, int64_t N, int64_t c0, int64_t W,
                                                              float* __restrict__ dpre, float* __restrict__ partial,
                                                              int64_t pstride, int rpb, const int64_t* __restrict__ rng_row,
                                                              bf16raw* __restrict__ o_hi, bf16raw* __restrict__ o_lo, int direct,
                                                              uint32_t* __restrict__ bits) {
  __shared__ __attribute__((aligned(16))) float red[1024];
  const RowColMap m = make_map(W);
  const bool drop = seed && p > 0.f;
  uint64_t key = 0;
  float inv = 1.f;
  if (drop) { key = rng_key(*seed, stream_id); inv = hw_rcp(1.f - p); }
  const float keep_scale = 1.f - p;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.active) {
    for (int r = m.r; r < rpb; r += m.rpp) {
      const int64_t row = r0 + r;
      if (row >= M) break;
      const int64_t off = row * N + c0 + m.c4 * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(dy + off);
      const float4 y4 = *reinterpret_cast<const float4*>(y + off);
      const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float f = 1.f, yy = yv[q];
        if (drop) {
          f = rng_keep(key, (uint64_t)((rng_row ? rng_row[row] * N + c0 + m.c4 * 4 : off) + q), p, inv);
          yy *= keep_scale;   // undo the 1/(1-p) on kept elements (dropped ones get f = 0 anyway)
        }
        o[q] = gv[q] * f * act_grad_from_out(act, yy);
      }
      if (dpre) *reinterpret_cast<float4*>(dpre + off) = make_float4(o[0], o[1], o[2], o[3]);     // (NULL: planes only)
      if (o_hi) {
        uint2 hh, ll;
        split4(make_float4(o[0], o[1], o[2], o[3]), hh, ll);
        *reinterpret_cast<uint2*>(o_hi + off) = hh;
        *reinterpret_cast<uint2*>(o_lo + off) = ll;
      }
      if (bits) {
        // (o > 0) as one bit per element, 32 columns per word: the 8 lanes of a word (host: W % 32 == 0, whole row groups per wave)
        // fold their nibbles with three exchanges. Consumed by the contraction epilogue's bit mask (advmil_epilogue_t.maskbits).
        uint32_t w = ((o[0] > 0.f) ? 1u : 0u) | ((o[1] > 0.f) ? 2u : 0u) | ((o[2] > 0.f) ? 4u : 0u) | ((o[3] > 0.f) ? 8u : 0u);
        w <<= 4 * (m.c4 & 7);
        w |= __shfl_xor(w, 1, 64); w |= __shfl_xor(w, 2, 64); w |= __shfl_xor(w, 4, 64);
        if ((m.c4 & 7) == 0) bits[row * (N >> 5) + ((c0 + m.c4 * 4) >> 5)] = w;
      }
      sum.x += o[0]; sum.y += o[1]; sum.z += o[2]; sum.w += o[3];
    }
  }
  if (partial) {
    float4 t = reduce_rows(sum, m, red);
    // direct (a single block covers every row: the [B, d] layers): `partial` IS the bias-gradient buffer (1: store, 2: add into)
    if (m.active && m.r == 0) {
      float4* dst = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * pstride + c0 + m.c4 * 4);
      if (direct == 2) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
      *dst = t;
    }
  }
}

extern "C" size_t advmil_colsum_workspace_bytes(int64_t M, int64_t N) {
  const int64_t nblk = (M + rows_per_block(M) - 1) / rows_per_block(M);
  return (size_t)(nblk * N) * sizeof(float);
}

extern "C" int advmil_act_dropout_bwd(const float* dy, const float* y, int act, float drop_p, const uint64_t* seed,
                                      uint64_t stream_id, int64_t M, int64_t N, float* dpre, float* dbias, int accumulate,
                                      const int64_t* rng_row, void* out_hi, void* out_lo, void* bits, void* ws, size_t ws_bytes,
                                      advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dy || !y || (!dpre && !out_hi) || M <= 0 || N <= 0 || (N & 3) || ((out_hi != nullptr) != (out_lo != nullptr))) return ADVMIL_EINVAL;
  // bit output: the 8 lanes of a word sit in one wave and one row when a row is a multiple of 32 columns (8 lanes, aligned to 8)
  if (bits && (N & 31)) return ADVMIL_EINVAL;
  if (dbias && (!ws || ws_bytes < advmil_colsum_workspace_bytes(M, N))) return ADVMIL_EWORKSPACE;
  const int rpb = rows_per_block(M);
  const int nblk = (int)((M + rpb - 1) / rpb);
  const bool direct = dbias && nblk == 1 && (((uintptr_t)dbias) & 15) == 0;      // one block sees every row: no partials, no merge launch
  float* partial = dbias ? (direct ? dbias : (float*)ws) : nullptr;
  for (int64_t c0 = 0; c0 < N; c0 += 1024) {
    const int64_t W = (N - c0 < 1024) ? (N - c0) : 1024;
    hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3(nblk), dim3(256), 0, stream, dy, y, act, drop_p, seed, stream_id, M, N,
                       c0, W, dpre, partial, N, rpb, rng_row, (bf16raw*)out_hi, (bf16raw*)out_lo, direct ? (accumulate ? 2 : 1) : 0,
                       (uint32_t*)bits);
  }
  ADVMIL_LAUNCH_CHECK();
  if (dbias && !direct) {
    const int rc = advmil_sumq(stream, partial, nblk, N, N, dbias, accumulate);
    if (rc) return rc;
  }
  return ADVMIL_OK;
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t M, int64_t N, int64_t c0,
                                                             int64_t W, float* __restrict__ partial, int rpb, int direct) {
  __shared__ __attribute__((aligned(16))) float red[1024];
  const RowColMap m = make_map(W);
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.active) {
    for (int r = m.r; r < rpb; r += m.rpp) {
      const int64_t row = r0 + r;
      if (row >= M) break;
      const float4 v = *reinterpret_cast<const float4*>(x + row * N + c0 + m.c4 * 4);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  float4 t = reduce_rows(sum, m, red);
  if (m.active && m.r == 0) {
    float4* dst = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * N + c0 + m.c4 * 4);
    if (direct == 2) { const float4 o = *dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    *dst = t;
  }
}

extern "C" int advmil_colsum(const float* x, int64_t M, int64_t N, float* out, int accumulate, void* ws, size_t ws_bytes,
                             advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || !ws || M <= 0 || N <= 0 || (N & 3)) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_colsum_workspace_bytes(M, N)) return ADVMIL_EWORKSPACE;
  const int rpb = rows_per_block(M);
  const int nblk = (int)((M + rpb - 1) / rpb);
  const bool direct = nblk == 1 && (((uintptr_t)out) & 15) == 0;
  for (int64_t c0 = 0; c0 < N; c0 += 1024) {
    const int64_t W = (N - c0 < 1024) ? (N - c0) : 1024;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, stream, x, M, N, c0, W, direct ? out : (float*)ws, rpb,
                       direct ? (accumulate ? 2 : 1) : 0);
  }
  ADVMIL_LAUNCH_CHECK();
  if (!direct) { const int rc = advmil_sumq(stream, (const float*)ws, nblk, N, N, out, accumulate); if (rc) return rc; }
  return ADVMIL_OK;
}

// =====================================================================================
// LayerNorm(d) -> ReLU -> mean over 16 consecutive rows.   One workgroup (4 waves) per region;
// wave w owns rows 4w..4w+3, lane owns columns lane, lane+64, ... (d <= 512).
// =====================================================================================
#define LN_MAXQ 8
// the region-embedding kernels are instantiated for the widths of the path (128: discriminator / PatchGCN, 256: GENConv MLP, 384:
// generator) so that a lane carries exactly d/64 columns; any other d <= 512 takes the 8-group instantiation
#define LN_DISPATCH(d, kernel, grid, stream, ...)                                                         \
  do {                                                                                                    \
    if ((d) <= 128) hipLaunchKernelGGL((kernel<2>), grid, dim3(256), 0, stream, __VA_ARGS__);             \
    else if ((d) <= 256) hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, stream, __VA_ARGS__);        \
    else if ((d) <= 384) hipLaunchKernelGGL((kernel<6>), grid, dim3(256), 0, stream, __VA_ARGS__);        \
    else hipLaunchKernelGGL((kernel<8>), grid, dim3(256), 0, stream, __VA_ARGS__);                        \
  } while (0)

template <int QT>
__global__ __launch_bounds__(256) void ln_relu_mean16_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, int64_t N,
                                                                 int64_t d, float* __restrict__ emb,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 int pool16) {
  __shared__ float red[4 * 512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t g = blockIdx.x;
  constexpr int Q = QT;      // 64-column groups per row (d <= 64 * QT)
  float gm[QT], bt[QT], acc[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    const int64_t j = lane + 64 * q;
    const bool ok = q < Q && j < d;
    gm[q] = ok ? gamma[j] : 0.f; bt[q] = ok ? beta[j] : 0.f; acc[q] = 0.f;
  }
  const float invd = hw_rcp((float)d);
  // the wave's four rows are independent: all their loads go out before the first reduction (one row at a time left a single
  // 256-byte request per wave in flight: 1.9 TB/s at d = 128)
  float va[4][QT];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = g * 16 + w * 4 + rr;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int64_t j = lane + 64 * q;
      va[rr][q] = (n < N && q < Q && j < d) ? y[n * d + j] : 0.f;
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = g * 16 + w * 4 + rr;
    if (n >= N) break;
    float (&v)[QT] = va[rr];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < QT; ++q) s += v[q];
    const float mu = wave_sum(s) * invd;
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int64_t j = lane + 64 * q;
      const float c = (q < Q && j < d) ? v[q] - mu : 0.f;
      s2 += c * c;
    }
    const float rs = hw_rsq(wave_sum(s2) * invd + eps);
    if (lane == 0) { mean[n] = mu; rstd[n] = rs; }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const float z = (v[q] - mu) * rs * gm[q] + bt[q];
      const float zr = z > 0.f ? z : 0.f;
      acc[q] += zr;
      const int64_t j = lane + 64 * q;
      if (!pool16 && q < Q && j < d) emb[n * d + j] = zr;          // plain LayerNorm+ReLU rows (GENConv MLP)
    }
  }
  if (!pool16) return;
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    const int64_t j = lane + 64 * q;
    if (q < Q && j < d) red[w * 512 + j] = acc[q];
  }
  __syncthreads();
  for (int64_t j = threadIdx.x; j < d; j += 256)
    emb[g * d + j] = (red[j] + red[512 + j] + red[1024 + j] + red[1536 + j]) * (1.f / 16.f);
}

// =====================================================================================
// LayerNorm -> ReLU -> mean16, 16-byte form (d % 128 == 0): a row is owned by HALF a wave, lane l of the half holds the float4 column
// groups 4 l + 128 c, c < NV = d / 128 -- every load / store is 16 bytes per lane (the operand planes 8), 512 contiguous bytes per row and
// instruction, two rows per wave-instruction. The 4-byte form above moved 256 (fp32) / 128 (planes) bytes per instruction: the backward
// over the [524288, 384] pre-activations of the ESAT 32k step ran at 2.1 TB/s (777 us), the d = 128 forms at 2.6-3.5 TB/s.
// One workgroup (4 waves) per 16-row region as before: wave w owns rows 4w .. 4w+3, half h = lane >> 5 rows 4w + h and 4w + 2 + h.
// =====================================================================================
__device__ __forceinline__ float half_sum(float v) {      // over the 32 lanes of a wave half
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NV>
__global__ __launch_bounds__(256) void ln_relu_mean16_fwd4_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, int64_t N, float* __restrict__ emb,
                                                                  float* __restrict__ mean, float* __restrict__ rstd,
                                                                  bf16raw* __restrict__ e_hi, bf16raw* __restrict__ e_lo, int dup) {
  constexpr int d = 128 * NV;
  __shared__ __attribute__((aligned(16))) float red[8][d];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l = lane & 31, h = lane >> 5;
  const int64_t g = blockIdx.x;
  float4 gm[NV], bt[NV], acc[NV], va[2][NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    gm[c] = *reinterpret_cast<const float4*>(gamma + c * 128 + 4 * l);
    bt[c] = *reinterpret_cast<const float4*>(beta + c * 128 + 4 * l);
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int64_t n = g * 16 + w * 4 + 2 * it + h;
#pragma unroll
    for (int c = 0; c < NV; ++c) va[it][c] = *reinterpret_cast<const float4*>(y + n * d + c * 128 + 4 * l);
  }
  const float invd = 1.0f / (float)d;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int64_t n = g * 16 + w * 4 + 2 * it + h;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) s += (va[it][c].x + va[it][c].y) + (va[it][c].z + va[it][c].w);
    const float mu = half_sum(s) * invd;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const float a0 = va[it][c].x - mu, a1 = va[it][c].y - mu, a2 = va[it][c].z - mu, a3 = va[it][c].w - mu;
      s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rs = hw_rsq(half_sum(s2) * invd + eps);
    if (l == 0) { mean[n] = mu; rstd[n] = rs; }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      acc[c].x += fmaxf((va[it][c].x - mu) * rs * gm[c].x + bt[c].x, 0.f);
      acc[c].y += fmaxf((va[it][c].y - mu) * rs * gm[c].y + bt[c].y, 0.f);
      acc[c].z += fmaxf((va[it][c].z - mu) * rs * gm[c].z + bt[c].z, 0.f);
      acc[c].w += fmaxf((va[it][c].w - mu) * rs * gm[c].w + bt[c].w, 0.f);
    }
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) *reinterpret_cast<float4*>(&red[w * 2 + h][c * 128 + 4 * l]) = acc[c];
  __syncthreads();
  for (int j = threadIdx.x; j < d / 4; j += 256) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(&red[r][4 * j]);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const float4 ev = make_float4(t.x * (1.f / 16.f), t.y * (1.f / 16.f), t.z * (1.f / 16.f), t.w * (1.f / 16.f));
    *reinterpret_cast<float4*>(emb + g * d + 4 * j) = ev;
    if (dup == 2) *reinterpret_cast<float4*>(emb + ((int64_t)gridDim.x + g) * d + 4 * j) = ev;      // [emb; emb]: the stacked fake | real pass
    if (e_hi) {                              // operand planes of the region embedding for the plane-fed in-projection behind it (ESAT)
      uint2 hh, ll;
      split4(ev, hh, ll);
      *reinterpret_cast<uint2*>(e_hi + g * d + 4 * j) = hh;
      *reinterpret_cast<uint2*>(e_lo + g * d + 4 * j) = ll;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256, (NV <= 2 ? 4 : 2)) void ln_relu_mean16_bwd4_kernel(const float* __restrict__ demb, const float* __restrict__ y,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd, int64_t N,
                                                                  float* __restrict__ dy, float* __restrict__ partial,
                                                                  bf16raw* __restrict__ o_hi, bf16raw* __restrict__ o_lo, int dup) {
  constexpr int d = 128 * NV;
  __shared__ __attribute__((aligned(16))) float red[8][3 * d];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l = lane & 31, h = lane >> 5;
  float4 gm[NV], bt[NV], ag[NV], abt[NV], ady[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    gm[c] = *reinterpret_cast<const float4*>(gamma + c * 128 + 4 * l);
    bt[c] = *reinterpret_cast<const float4*>(beta + c * 128 + 4 * l);
    ag[c] = abt[c] = ady[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invd = 1.0f / (float)d;
  const int64_t nreg = N / 16;
  for (int64_t g = blockIdx.x; g < nreg; g += gridDim.x) {
    float4 de[NV], ya[2][NV];
    float mua[2], rsa[2];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      float4 v = *reinterpret_cast<const float4*>(demb + g * d + c * 128 + 4 * l);
      if (dup == 2) {        // the embedding fed two stacked passes: its gradient is the sum of the two halves of demb [2 nreg, d]
        const float4 v2 = *reinterpret_cast<const float4*>(demb + (nreg + g) * d + c * 128 + 4 * l);
        v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
      }
      de[c] = make_float4(v.x * (1.f / 16.f), v.y * (1.f / 16.f), v.z * (1.f / 16.f), v.w * (1.f / 16.f));
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int64_t n = g * 16 + w * 4 + 2 * it + h;
      mua[it] = mean[n]; rsa[it] = rstd[n];
#pragma unroll
      for (int c = 0; c < NV; ++c) ya[it][c] = *reinterpret_cast<const float4*>(y + n * d + c * 128 + 4 * l);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int64_t n = g * 16 + w * 4 + 2 * it + h;
      const float mu = mua[it], rs = rsa[it];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const float yv[4] = {ya[it][c].x, ya[it][c].y, ya[it][c].z, ya[it][c].w};
        const float gv[4] = {gm[c].x, gm[c].y, gm[c].z, gm[c].w}, bv[4] = {bt[c].x, bt[c].y, bt[c].z, bt[c].w};
        const float dv[4] = {de[c].x, de[c].y, de[c].z, de[c].w};
        float a4[4], b4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float xo = (yv[t] - mu) * rs;
          const float dz = (xo * gv[t] + bv[t]) > 0.f ? dv[t] : 0.f;
          const float dxo = dz * gv[t];
          a4[t] = dz * xo;
          b4[t] = dz;
          c1 += dxo;
          c2 += dxo * xo;
        }
        ag[c].x += a4[0]; ag[c].y += a4[1]; ag[c].z += a4[2]; ag[c].w += a4[3];
        abt[c].x += b4[0]; abt[c].y += b4[1]; abt[c].z += b4[2]; abt[c].w += b4[3];
      }
      c1 = half_sum(c1) * invd;
      c2 = half_sum(c2) * invd;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        // (x-hat and dz * gamma recomputed instead of carried through the two reductions: registers that decide the waves per SIMD)
        const float yv[4] = {ya[it][c].x, ya[it][c].y, ya[it][c].z, ya[it][c].w};
        const float gv[4] = {gm[c].x, gm[c].y, gm[c].z, gm[c].w}, bv[4] = {bt[c].x, bt[c].y, bt[c].z, bt[c].w};
        const float dv[4] = {de[c].x, de[c].y, de[c].z, de[c].w};
        float o4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float xo = (yv[t] - mu) * rs;
          const float dxo = ((xo * gv[t] + bv[t]) > 0.f ? dv[t] : 0.f) * gv[t];
          o4[t] = rs * (dxo - c1 - xo * c2);
        }
        const float4 v = make_float4(o4[0], o4[1], o4[2], o4[3]);
        const int64_t o = n * d + c * 128 + 4 * l;
        if (dy) *reinterpret_cast<float4*>(dy + o) = v;
        if (o_hi) {
          uint2 hh, ll;
          split4(v, hh, ll);
          *reinterpret_cast<uint2*>(o_hi + o) = hh;
          *reinterpret_cast<uint2*>(o_lo + o) = ll;
        }
        ady[c].x += v.x; ady[c].y += v.y; ady[c].z += v.z; ady[c].w += v.w;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    *reinterpret_cast<float4*>(&red[w * 2 + h][c * 128 + 4 * l]) = ag[c];
    *reinterpret_cast<float4*>(&red[w * 2 + h][d + c * 128 + 4 * l]) = abt[c];
    *reinterpret_cast<float4*>(&red[w * 2 + h][2 * d + c * 128 + 4 * l]) = ady[c];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 3 * d / 4; j += 256) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(&red[r][4 * j]);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * 3 * d + 4 * j) = t;
  }
}

#define LN4_DISPATCH(d, kernel, grid, stream, ...)                                                       \
  do {                                                                                                   \
    switch ((int)((d) / 128)) {                                                                          \
      case 1: hipLaunchKernelGGL((kernel<1>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
      case 2: hipLaunchKernelGGL((kernel<2>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
      case 3: hipLaunchKernelGGL((kernel<3>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
      default: hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, stream, __VA_ARGS__); break;          \
    }                                                                                                    \
  } while (0)

static const bool g_ln4 = []() { const char* e = getenv("ADVMIL_LN4"); return !(e && e[0] == '0'); }();

extern "C" int advmil_ln_relu_mean16_fwd(const float* y, const float* gamma, const float* beta, float eps, int64_t N,
                                         int64_t d, float* emb, float* mean, float* rstd, void* emb_hi, void* emb_lo, int dup,
                                         advmil_stream_t stream) {
  if (!y || !gamma || !beta || !emb || !mean || !rstd || N <= 0 || (N & 15) || d <= 0 || d > 512) return ADVMIL_EINVAL;
  if ((emb_hi != nullptr) != (emb_lo != nullptr) || (dup != 1 && dup != 2)) return ADVMIL_EINVAL;
  if (g_ln4 && (d & 127) == 0 && !((((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)emb)) & 15) &&
      !((((uintptr_t)emb_hi) | ((uintptr_t)emb_lo)) & 7)) {
    LN4_DISPATCH(d, ln_relu_mean16_fwd4_kernel, dim3((unsigned)(N / 16)), (hipStream_t)stream, y, gamma, beta, eps, N, emb, mean, rstd,
                 (bf16raw*)emb_hi, (bf16raw*)emb_lo, dup);
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  if (emb_hi || dup != 1) return ADVMIL_EINVAL;      // (plane output and duplication ride in the 16-byte form only: d % 128 == 0)
  LN_DISPATCH(d, ln_relu_mean16_fwd_kernel, dim3((unsigned)(N / 16)), (hipStream_t)stream, y, gamma, beta, eps, N, d, emb, mean, rstd, 1);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_ln_relu_fwd(const float* y, const float* gamma, const float* beta, float eps, int64_t N, int64_t d,
                                  float* out, float* mean, float* rstd, advmil_stream_t stream) {
  if (!y || !gamma || !beta || !out || !mean || !rstd || N <= 0 || d <= 0 || d > 512) return ADVMIL_EINVAL;
  LN_DISPATCH(d, ln_relu_mean16_fwd_kernel, dim3((unsigned)((N + 15) / 16)), (hipStream_t)stream, y, gamma, beta, eps, N, d, out, mean, rstd, 0);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

template <int QT>
__global__ __launch_bounds__(256, (QT <= 6 ? 4 : 2)) void ln_relu_mean16_bwd_kernel(const float* __restrict__ demb, const float* __restrict__ y,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 int64_t N, int64_t d, float* __restrict__ dy,
                                                                 float* __restrict__ partial, int pool16,
                                                                 bf16raw* __restrict__ o_hi, bf16raw* __restrict__ o_lo) {
  __shared__ float red[4 * 1536];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int Q = QT;      // 64-column groups per row (d <= 64 * QT)
  float gm[QT], bt[QT], de[QT], ag[QT], abt[QT], ady[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    const int64_t j = lane + 64 * q;
    const bool ok = q < Q && j < d;
    gm[q] = ok ? gamma[j] : 0.f; bt[q] = ok ? beta[j] : 0.f;
    de[q] = 0.f; ag[q] = 0.f; abt[q] = 0.f; ady[q] = 0.f;
  }
  const float invd = hw_rcp((float)d);
  const int64_t nreg = (N + 15) / 16;
  // a workgroup walks the 16-row regions blockIdx.x, + gridDim.x, ... and leaves ONE row of gamma / beta partials: the merge then
  // reads <= 2048 rows whatever the slab size (a 32768-patch step slab used to leave 32768 of them)
  for (int64_t g = blockIdx.x; g < nreg; g += gridDim.x) {
  if (pool16) {
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int64_t j = lane + 64 * q;
      de[q] = (q < Q && j < d) ? demb[g * d + j] * (1.f / 16.f) : 0.f;
    }
  }
  // the wave's four rows are independent: all their loads go out before the first reduction
  float ya[4][QT], mua[4], rsa[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = g * 16 + w * 4 + rr;
    mua[rr] = n < N ? mean[n] : 0.f;
    rsa[rr] = n < N ? rstd[n] : 0.f;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int64_t j = lane + 64 * q;
      ya[rr][q] = (n < N && q < Q && j < d) ? y[n * d + j] : 0.f;
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = g * 16 + w * 4 + rr;
    if (n >= N) break;
    const float mu = mua[rr], rs = rsa[rr];
    if (!pool16) {
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        const int64_t j = lane + 64 * q;
        de[q] = (q < Q && j < d) ? demb[n * d + j] : 0.f;
      }
    }
    float xh[QT];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int64_t j = lane + 64 * q;
      const bool ok = q < Q && j < d;
      xh[q] = ok ? (ya[rr][q] - mu) * rs : 0.f;
      const float z = xh[q] * gm[q] + bt[q];
      const float dz = (ok && z > 0.f) ? de[q] : 0.f;
      const float dxh = dz * gm[q];
      ag[q] += dz * xh[q];
      abt[q] += dz;
      c1 += dxh;
      c2 += dxh * xh[q];
    }
    c1 = wave_sum(c1) * invd;
    c2 = wave_sum(c2) * invd;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int64_t j = lane + 64 * q;
      if (q < Q && j < d) {
        // (dz * gamma recomputed instead of carried through the two reductions: six registers that decide 3 vs 4 waves per SIMD)
        const float dxh = ((xh[q] * gm[q] + bt[q]) > 0.f ? de[q] : 0.f) * gm[q];
        const float v = rs * (dxh - c1 - xh[q] * c2);
        if (dy) dy[n * d + j] = v;                        // (NULL: planes only)
        if (o_hi) {                                       // bf16x3 operand planes of dy for the weight-gradient contraction that reads it
          unsigned hh, ll;
          split2(v, 0.f, hh, ll);
          o_hi[n * d + j] = (bf16raw)(hh & 0xffffu);
          o_lo[n * d + j] = (bf16raw)(ll & 0xffffu);
        }
        ady[q] += v;            // column sums of dy = the bias gradient of the layer that produced y (no second pass over dy)
      }
    }
  }
  }
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    const int64_t j = lane + 64 * q;
    if (q < Q && j < d) { red[w * 1536 + j] = ag[q]; red[w * 1536 + 512 + j] = abt[q]; red[w * 1536 + 1024 + j] = ady[q]; }
  }
  __syncthreads();
  const int64_t gp = blockIdx.x;
  for (int64_t j = threadIdx.x; j < d; j += 256) {
    partial[gp * 3 * d + j] = red[j] + red[1536 + j] + red[3072 + j] + red[4608 + j];
    partial[gp * 3 * d + d + j] = red[512 + j] + red[2048 + j] + red[3584 + j] + red[5120 + j];
    partial[gp * 3 * d + 2 * d + j] = red[1024 + j] + red[2560 + j] + red[4096 + j] + red[5632 + j];
  }
}

#define LN_BWD_MAXBLK 2048
static inline int ln_bwd_blocks(int64_t nreg) { return (int)(nreg < LN_BWD_MAXBLK ? nreg : LN_BWD_MAXBLK); }

extern "C" size_t advmil_ln_relu_mean16_bwd_workspace_bytes(int64_t N, int64_t d) {
  return (size_t)(ln_bwd_blocks(N / 16) * 3 * d) * sizeof(float);
}

extern "C" int advmil_ln_relu_mean16_bwd(const float* demb, const float* y, const float* gamma, const float* beta,
                                         const float* mean, const float* rstd, int64_t N, int64_t d, float* dy,
                                         float* dgamma, float* dbeta, int accumulate, float* dycol, void* dy_hi, void* dy_lo, int dup,
                                         void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!demb || !y || !gamma || !beta || !mean || !rstd || (!dy && !dy_hi) || !dgamma || !dbeta || !ws || N <= 0 || (N & 15) || d <= 0 ||
      d > 512 || ((dy_hi != nullptr) != (dy_lo != nullptr)))
    return ADVMIL_EINVAL;
  if (ws_bytes < advmil_ln_relu_mean16_bwd_workspace_bytes(N, d)) return ADVMIL_EWORKSPACE;
  const int L = ln_bwd_blocks(N / 16);
  float* partial = (float*)ws;
  if (g_ln4 && (d & 127) == 0 && !((((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)demb) | ((uintptr_t)dy) | ((uintptr_t)partial)) & 15) &&
      !((((uintptr_t)dy_hi) | ((uintptr_t)dy_lo)) & 7)) {
    LN4_DISPATCH(d, ln_relu_mean16_bwd4_kernel, dim3(L), stream, demb, y, gamma, beta, mean, rstd, N, dy, partial, (bf16raw*)dy_hi, (bf16raw*)dy_lo, dup);
  } else if (dup != 1) {
    return ADVMIL_EINVAL;
  } else
  LN_DISPATCH(d, ln_relu_mean16_bwd_kernel, dim3(L), stream, demb, y, gamma, beta, mean, rstd, N, d, dy, partial, 1, (bf16raw*)dy_hi,
              (bf16raw*)dy_lo);
  ADVMIL_LAUNCH_CHECK();
  // dgamma | dbeta | column sums of dy are adjacent in the partial rows: ONE merge entry with three destinations when all three accumulate
  // (dycol -- the bias gradient of the FC that produced y -- always does: it is the caller's accumulator)
  int rc;
  if (dycol && accumulate) {
    rc = advmil_sumq(stream, partial, L, 3 * d, 3 * d, dgamma, 1, dbeta, d, dycol, 2 * d);
  } else {
    rc = advmil_sumq(stream, partial, L, 3 * d, 2 * d, dgamma, accumulate, dbeta, d);
    if (!rc && dycol) rc = advmil_sumq(stream, partial + 2 * d, L, 3 * d, d, dycol, 1);
  }
  return rc;
}

extern "C" size_t advmil_ln_relu_bwd_workspace_bytes(int64_t N, int64_t d) {
  return (size_t)(ln_bwd_blocks((N + 15) / 16) * 3 * d) * sizeof(float);
}

extern "C" int advmil_ln_relu_bwd(const float* dout, const float* y, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, int64_t N, int64_t d, float* dy, float* dgamma, float* dbeta,
                                  int accumulate, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dout || !y || !gamma || !beta || !mean || !rstd || !dy || !dgamma || !dbeta || !ws || N <= 0 || d <= 0 || d > 512)
    return ADVMIL_EINVAL;
  if (ws_bytes < advmil_ln_relu_bwd_workspace_bytes(N, d)) return ADVMIL_EWORKSPACE;
  const int L = ln_bwd_blocks((N + 15) / 16);
  float* partial = (float*)ws;
  LN_DISPATCH(d, ln_relu_mean16_bwd_kernel, dim3(L), stream, dout, y, gamma, beta, mean, rstd, N, d, dy, partial, 0, (bf16raw*)nullptr,
              (bf16raw*)nullptr);
  ADVMIL_LAUNCH_CHECK();
  return advmil_sumq(stream, partial, L, 3 * d, 2 * d, dgamma, accumulate, dbeta, d);
}

// =====================================================================================
// Post-norm residual of the ESAT transformer layer (nn.TransformerEncoderLayer, norm_first = False; reference
// model/backbone_utils.py:113-127):   y = LayerNorm(x + dropout(o))   one wave per row, lane owns columns lane, lane+64, ...
//   fwd also stores z = x + dropout(o) and the row statistics for the backward; dropout index = row*d + col on `stream_id`
//   (the flat index of advmil_dropout_apply, so the host regenerates the mask the same way).
//   bwd: dz = LayerNorm'(dy); dx = dz; do = dz * keep; dgamma += sum_rows dy*xhat; dbeta += sum_rows dy.
// =====================================================================================
__global__ __launch_bounds__(256) void add_dropout_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ o,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, int64_t R, int64_t d, float p, const uint64_t* seed,
                                                                 uint64_t stream_id, float* __restrict__ z, float* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 const int64_t* __restrict__ rng_row, bf16raw* __restrict__ y_hi,
                                                                 bf16raw* __restrict__ y_lo) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int Q = (int)((d + 63) / 64);
  const bool drop = seed && p > 0.f;
  uint64_t key = 0;
  float ik = 1.f;
  if (drop) { key = rng_key(*seed, stream_id); ik = hw_rcp(1.f - p); }
  const float invd = hw_rcp((float)d);
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = (int64_t)blockIdx.x * 16 + w * 4 + rr;
    if (n >= R) break;
    float v[LN_MAXQ];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      const int64_t j = lane + 64 * q;
      const bool ok = q < Q && j < d;
      float t = 0.f;
      if (ok) {
        float ov = o[n * d + j];
        if (drop) ov *= rng_keep(key, (uint64_t)((rng_row ? rng_row[n] : n) * d + j), p, ik);
        t = x[n * d + j] + ov;
        z[n * d + j] = t;
      }
      v[q] = t;
      s += t;
    }
    const float mu = wave_sum(s) * invd;
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      const int64_t j = lane + 64 * q;
      const float c = (q < Q && j < d) ? v[q] - mu : 0.f;
      s2 += c * c;
    }
    const float rs = hw_rsq(wave_sum(s2) * invd + eps);
    if (lane == 0) { mean[n] = mu; rstd[n] = rs; }
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      const int64_t j = lane + 64 * q;
      if (q < Q && j < d) {
        const float yv = (v[q] - mu) * rs * gamma[j] + beta[j];
        y[n * d + j] = yv;
        if (y_hi) {                          // operand planes of y for the plane-fed contraction that reads it next (FFN / gate branches)
          unsigned hh, ll;
          split2(yv, 0.f, hh, ll);
          y_hi[n * d + j] = (bf16raw)(hh & 0xffffu);
          y_lo[n * d + j] = (bf16raw)(ll & 0xffffu);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void add_dropout_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int64_t R, int64_t d, float p,
                                                                 const uint64_t* seed, uint64_t stream_id, float* __restrict__ dx,
                                                                 float* __restrict__ dob, float* __restrict__ partial,
                                                                 const int64_t* __restrict__ rng_row) {
  __shared__ float red[4 * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t g = blockIdx.x;
  const int Q = (int)((d + 63) / 64);
  const bool drop = seed && p > 0.f;
  uint64_t key = 0;
  float ik = 1.f;
  if (drop) { key = rng_key(*seed, stream_id); ik = hw_rcp(1.f - p); }
  float gm[LN_MAXQ], ag[LN_MAXQ], abt[LN_MAXQ];
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) {
    const int64_t j = lane + 64 * q;
    gm[q] = (q < Q && j < d) ? gamma[j] : 0.f;
    ag[q] = 0.f; abt[q] = 0.f;
  }
  const float invd = hw_rcp((float)d);
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t n = g * 16 + w * 4 + rr;
    if (n >= R) break;
    const float mu = mean[n], rs = rstd[n];
    float xh[LN_MAXQ], dxh[LN_MAXQ];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      const int64_t j = lane + 64 * q;
      const bool ok = q < Q && j < d;
      xh[q] = ok ? (z[n * d + j] - mu) * rs : 0.f;
      const float g_ = ok ? dy[n * d + j] : 0.f;
      dxh[q] = g_ * gm[q];
      ag[q] += g_ * xh[q];
      abt[q] += g_;
      c1 += dxh[q];
      c2 += dxh[q] * xh[q];
    }
    c1 = wave_sum(c1) * invd;
    c2 = wave_sum(c2) * invd;
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      const int64_t j = lane + 64 * q;
      if (q < Q && j < d) {
        const float dz = rs * (dxh[q] - c1 - xh[q] * c2);
        dx[n * d + j] = dz;
        if (dob) dob[n * d + j] = drop ? dz * rng_keep(key, (uint64_t)((rng_row ? rng_row[n] : n) * d + j), p, ik) : dz;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) {
    const int64_t j = lane + 64 * q;
    if (q < Q && j < d) { red[w * 1024 + j] = ag[q]; red[w * 1024 + 512 + j] = abt[q]; }
  }
  __syncthreads();
  for (int64_t j = threadIdx.x; j < d; j += 256) {
    partial[g * 2 * d + j] = red[j] + red[1024 + j] + red[2048 + j] + red[3072 + j];
    partial[g * 2 * d + d + j] = red[512 + j] + red[1536 + j] + red[2560 + j] + red[3584 + j];
  }
}

extern "C" int advmil_add_dropout_ln_fwd(const float* x, const float* o, const float* gamma, const float* beta, float eps,
                                         int64_t R, int64_t d, float drop_p, const uint64_t* seed, uint64_t stream_id,
                                         const int64_t* rng_row, float* z, float* y, float* mean, float* rstd, void* y_hi, void* y_lo,
                                         advmil_stream_t stream) {
  if (!x || !o || !gamma || !beta || !z || !y || !mean || !rstd || R <= 0 || d <= 0 || d > 512) return ADVMIL_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || ((y_hi != nullptr) != (y_lo != nullptr))) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(add_dropout_ln_fwd_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, o, gamma,
                     beta, eps, R, d, drop_p, seed, stream_id, z, y, mean, rstd, rng_row, (bf16raw*)y_hi, (bf16raw*)y_lo);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" size_t advmil_add_dropout_ln_bwd_workspace_bytes(int64_t R, int64_t d) {
  return (size_t)(((R + 15) / 16) * 2 * d) * sizeof(float);
}

extern "C" int advmil_add_dropout_ln_bwd(const float* dy, const float* z, const float* gamma, const float* mean, const float* rstd,
                                         int64_t R, int64_t d, float drop_p, const uint64_t* seed, uint64_t stream_id,
                                         const int64_t* rng_row, float* dx, float* dob, float* dgamma, float* dbeta,
                                         int accumulate, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dy || !z || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !ws || R <= 0 || d <= 0 || d > 512) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_add_dropout_ln_bwd_workspace_bytes(R, d)) return ADVMIL_EWORKSPACE;
  const int nb = (int)((R + 15) / 16);
  float* partial = (float*)ws;
  hipLaunchKernelGGL(add_dropout_ln_bwd_kernel, dim3(nb), dim3(256), 0, stream, dy, z, gamma, mean, rstd, R, d, drop_p, seed,
                     stream_id, dx, dob, partial, rng_row);
  ADVMIL_LAUNCH_CHECK();
  return advmil_sumq(stream, partial, nb, 2 * d, 2 * d, dgamma, accumulate, dbeta, d);
}
