// K6: GENConv softmax aggregation of PatchGCN (model/backbone.py:139,157; arithmetic of torch_geometric.nn.GENConv,
// restated from its published semantics -- parity unpinned, see oracle/advmil_oracle.py::genconv):
//   message  m_j   = relu(x_j) + eps                                  (per source node j, per channel c)
//   weights  w_ij  = softmax over the in-edges j->i of (t * m_j)      (per target i, per channel c)
//   output   out_i = agg_i + x_i,  agg_i = sum_j w_ij m_j
//   backward dx_j  = dout_j + relu'(x_j) * sum_{edges j->i} dout_i * w_ij * (1 + t (m_j - agg_i)),   w_ij = exp(t m_j - lse_i)
//            dt    = sum_{edges j->i, c} dout_i * w_ij * m_j * (m_j - agg_i)        (= sum_i dout_i (E_w[m^2] - agg_i^2))
// HBM-bound sparse gather: neighbours come from a CSR list, lanes own channels, so every row read is a run of full lines and there
// are no atomics: the forward walks the graph by DESTINATION, the backward by SOURCE (both CSR images are built once per graph on
// the host side). The forward keeps lse and agg per node (two rows) so the backward's edge walk gathers THREE rows per edge
// (dout, lse, agg of the target) and nothing is re-derived from out - x; dt falls out of the same walk (one partial per workgroup,
// summed in a fixed order by a one-workgroup kernel: run-to-run identical), so no pass over the node arrays is left on the caller's side.
// Workgroup -> node-tile mapping: workgroups are dealt to the 8 XCDs round-robin (b % 8) and each XCD has its own L2, so tile
// (b % 8) * per_xcd + b / 8 gives every XCD ONE contiguous eighth of the nodes: a neighbour row fetched by one workgroup is an L2 hit
// for the workgroups around it (k-NN graphs over patch coordinates are local in node order), instead of being fetched into all eight L2s.
#include "common.h"
#include "../../include/advmil_hip.h"

#ifndef GENCONV_BWD_CH
#define GENCONV_BWD_CH 4      // out-edges of a source node whose three rows are in flight together (backward)
#endif

__device__ __forceinline__ int64_t xcd_tile(unsigned b, unsigned per_xcd) { return (int64_t)(b & 7u) * per_xcd + (b >> 3); }

// partial of dt for this workgroup (256 threads): every thread calls it
__device__ __forceinline__ void dt_partial_store(float gt, float* __restrict__ dt_part) {
  __shared__ float red[4];
  const float s = wave_sum(gt);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dt_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void genconv_dt_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ dt) {
  __shared__ float red[256];
  float s = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) s += part[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dt[0] = red[0];
}

// ---- any C: one wave per node, lanes stride over the channels
__global__ __launch_bounds__(256) void genconv_fwd_kernel(const float* __restrict__ x, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col, const float* __restrict__ tptr,
                                                          float eps, int64_t N, int64_t C, unsigned per_xcd, float* __restrict__ out,
                                                          float* __restrict__ lse, float* __restrict__ agg_out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = xcd_tile(blockIdx.x, per_xcd) * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float t = tptr[0];
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int64_t c = lane; c < C; c += 64) {
    float mx = -INFINITY;
    for (int e = e0; e < e1; ++e) {
      const float v = x[(int64_t)col[e] * C + c];
      mx = fmaxf(mx, t * ((v > 0.f ? v : 0.f) + eps));
    }
    float den = 0.f, a1 = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float v = x[(int64_t)col[e] * C + c];
      const float m = (v > 0.f ? v : 0.f) + eps;
      const float w = hw_exp(t * m - mx);
      den += w; a1 += w * m;
    }
    const bool has = e1 > e0;
    const float agg = has ? a1 * hw_rcp(den) : 0.f;
    out[i * C + c] = agg + x[i * C + c];
    if (lse) {
      lse[i * C + c] = has ? (mx + hw_log(den)) * 1.44269504088896340736f : 0.f;      // saved in log2 units (see the C == 128 kernels)
      agg_out[i * C + c] = agg;
    }
  }
}

__global__ __launch_bounds__(256) void genconv_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                          const float* __restrict__ agg, const float* __restrict__ lse,
                                                          const int* __restrict__ rowptr_s, const int* __restrict__ col_s,
                                                          const float* __restrict__ tptr, float eps, int64_t N, int64_t C,
                                                          unsigned per_xcd, float* __restrict__ dx, float* __restrict__ dt_part) {
  const int lane = threadIdx.x & 63;
  const int64_t j = xcd_tile(blockIdx.x, per_xcd) * 4 + (threadIdx.x >> 6);
  float gt = 0.f;
  if (j < N) {
    const float t = tptr[0];
    const int e0 = rowptr_s[j], e1 = rowptr_s[j + 1];
    for (int64_t c = lane; c < C; c += 64) {
      const float xj = x[j * C + c];
      const float m = (xj > 0.f ? xj : 0.f) + eps;
      float g = 0.f;
      for (int e = e0; e < e1; ++e) {
        const int64_t i = col_s[e];
        const float dw = dout[i * C + c] * hw_exp2(t * 1.44269504088896340736f * m - lse[i * C + c]);
        const float dm = m - agg[i * C + c];
        g += dw * (1.f + t * dm);
        gt += dw * m * dm;
      }
      dx[j * C + c] = dout[j * C + c] + (xj > 0.f ? g : 0.f);
    }
  }
  dt_partial_store(gt, dt_part);
}

// ---- C == 128 (the PatchGCN width, model/backbone.py:40): a 32-lane half-wave owns a node, one float4 (16 B) per lane covers its
// 512-byte row, 8 consecutive nodes per workgroup (their neighbourhoods overlap: ~40 % of the gathered lines hit in the CU's L1); the
// neighbour rows are gathered ONCE, a chunk of edges at a time with every index and every row of the chunk in flight together (8 rows in
// the forward = the k-NN degree of tools/patchgcn_graph_s2.py, CH x 3 rows in the backward); longer lists are merged chunk by chunk
// (online-softmax rescale in the forward).
// Measured (profiles/r04_pmc_genconv.txt): memory-side reads and writes are the compulsory bytes (FETCH_SIZE x 2 = 37 / 139 MB, WRITE_SIZE
// 98 / 33 MB), the L2 hit rate is 0.78 on reads; what is busy is the CU's texture-address / L1 path (TA busy 0.72-0.76 of the launch at one
// 64-B access per ~2 clocks): the kernels are bound by the bytes that pass through L1 (301 / 872 MB gathered), not by HBM. Staging the
// row pointers and edge lists of a 32-node tile in LDS (no dependent rowptr -> col -> row chain) measured no faster (45 / 70 us vs 41 / 69).
// Arithmetic: the forward is VALU-bound, not memory-bound (one exp per edge and channel: ~18 issue slots per gathered element as first
// written = 35 us of the 40 us launch), so the inner loops are written for issue slots: exponents live in the log2 domain
// (tl = t log2 e; `lse` is SAVED in log2 units and only ever read back by the backward), the running maximum is taken over the messages
// themselves with v_max3 (t m is monotone in m: max for t >= 0, min for t < 0 -- a wave-uniform branch), full chunks of 8 edges run
// without per-edge predicates, and four exponentials share one trans -> VALU hazard pad.
__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int64_t row, int l) {
  return *reinterpret_cast<const float4*>(p + row * 128 + 4 * l);
}
__device__ __forceinline__ float relu_eps(float v, float eps) {      // (plain v_max: fmaxf() costs a canonicalising v_max per operand)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r + eps;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ void exp2x4(const float (&e)[4], float (&w)[4]) {
  asm("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\ts_nop 1"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
}
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.693147180559945309f;

// one chunk of up to 8 in-edges of a node, four channels per lane. mm = running extremum of the messages, mx = tl * mm (log2 units).
template <bool FULL, bool POS>
__device__ __forceinline__ void fwd_chunk(const float* __restrict__ x, const int* __restrict__ col, int eb, int n, int l, float eps,
                                          float tl, bool first, float (&mm)[4], float (&mx)[4], float (&den)[4], float (&a1)[4]) {
  int idx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) idx[k] = col[eb + ((FULL || k < n) ? k : 0)];      // past the list: the chunk's first edge again
  float m[8][4];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float4 v = ld4(x, idx[k], l);
    m[k][0] = relu_eps(v.x, eps); m[k][1] = relu_eps(v.y, eps); m[k][2] = relu_eps(v.z, eps); m[k][3] = relu_eps(v.w, eps);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {                     // (a repeated row does not move an extremum: no predicate needed here)
#pragma unroll
    for (int k = 0; k < 8; k += 2) mm[q] = POS ? max3f(mm[q], m[k][q], m[k + 1][q]) : min3f(mm[q], m[k][q], m[k + 1][q]);
  }
  float cm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cm[q] = tl * mm[q];
  if (!first) {
    float e[4], sc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = mx[q] - cm[q];
    exp2x4(e, sc);
#pragma unroll
    for (int q = 0; q < 4; ++q) { den[q] *= sc[q]; a1[q] *= sc[q]; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) mx[q] = cm[q];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (FULL || k < n) {
      float e[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = fmaf(m[k][q], tl, -mx[q]);
      exp2x4(e, w);
#pragma unroll
      for (int q = 0; q < 4; ++q) { den[q] += w[q]; a1[q] = fmaf(w[q], m[k][q], a1[q]); }
    }
  }
}

template <bool SAVE>
__global__ __launch_bounds__(256) void genconv_fwd128_kernel(const float* __restrict__ x, const int* __restrict__ rowptr,
                                                             const int* __restrict__ col, const float* __restrict__ tptr, float eps,
                                                             int64_t N, unsigned per_xcd, float* __restrict__ out,
                                                             float* __restrict__ lse, float* __restrict__ agg_out) {
  constexpr int C = 128;
  const int l = threadIdx.x & 31;
  const int64_t i = xcd_tile(blockIdx.x, per_xcd) * 8 + (threadIdx.x >> 5);
  if (i >= N) return;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  const float tl = tptr[0] * LOG2E;
  const float4 xi = ld4(x, i, l);
  float mm[4], mx[4] = {0.f, 0.f, 0.f, 0.f}, den[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
  if (tl >= 0.f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) mm[q] = -INFINITY;
    for (int eb = e0; eb < e1; eb += 8) {
      if (e1 - eb >= 8) fwd_chunk<true, true>(x, col, eb, 8, l, eps, tl, eb == e0, mm, mx, den, a1);
      else fwd_chunk<false, true>(x, col, eb, e1 - eb, l, eps, tl, eb == e0, mm, mx, den, a1);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) mm[q] = INFINITY;
    for (int eb = e0; eb < e1; eb += 8) fwd_chunk<false, false>(x, col, eb, (e1 - eb) < 8 ? (e1 - eb) : 8, l, eps, tl, eb == e0, mm, mx, den, a1);
  }
  const bool has = e1 > e0;
  const float xv[4] = {xi.x, xi.y, xi.z, xi.w};
  float o[4], ls[4], ag[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ag[q] = has ? a1[q] * hw_rcp(den[q]) : 0.f;
    o[q] = ag[q] + xv[q];
    ls[q] = has ? mx[q] + hw_log2(den[q]) : 0.f;
  }
  *reinterpret_cast<float4*>(out + i * C + 4 * l) = make_float4(o[0], o[1], o[2], o[3]);
  if (SAVE) {
    *reinterpret_cast<float4*>(lse + i * C + 4 * l) = make_float4(ls[0], ls[1], ls[2], ls[3]);
    *reinterpret_cast<float4*>(agg_out + i * C + 4 * l) = make_float4(ag[0], ag[1], ag[2], ag[3]);
  }
}

// backward, per source node j and channel:  g1 = sum_i dw_i,  g2 = sum_i dw_i (m_j - agg_i),  dw_i = dout_i exp2(tl m_j - lse2_i)
//   dx_j = dout_j + relu'(x_j) (g1 + t g2),   dt += m_j g2
template <int CH, bool FULL>
__device__ __forceinline__ void bwd_chunk(const float* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ agg,
                                          const int* __restrict__ col_s, int eb, int n, int l, const float (&m)[4],
                                          const float (&tm)[4], float (&g1)[4], float (&g2)[4]) {
  int idx[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) idx[k] = col_s[eb + ((FULL || k < n) ? k : 0)];
  float4 d4[CH], l4[CH], a4[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) { d4[k] = ld4(dout, idx[k], l); l4[k] = ld4(lse, idx[k], l); a4[k] = ld4(agg, idx[k], l); }
#pragma unroll
  for (int k = 0; k < CH; ++k)
    if (FULL || k < n) {
      const float dv[4] = {d4[k].x, d4[k].y, d4[k].z, d4[k].w}, lv[4] = {l4[k].x, l4[k].y, l4[k].z, l4[k].w},
                  av[4] = {a4[k].x, a4[k].y, a4[k].z, a4[k].w};
      float e[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = tm[q] - lv[q];
      exp2x4(e, w);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dw = dv[q] * w[q];
        g1[q] += dw;
        g2[q] = fmaf(dw, m[q] - av[q], g2[q]);
      }
    }
}

template <int CH>
__global__ __launch_bounds__(256) void genconv_bwd128_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                             const float* __restrict__ agg, const float* __restrict__ lse,
                                                             const int* __restrict__ rowptr_s, const int* __restrict__ col_s,
                                                             const float* __restrict__ tptr, float eps, int64_t N, unsigned per_xcd,
                                                             float* __restrict__ dx, float* __restrict__ dt_part) {
  constexpr int C = 128;
  const int l = threadIdx.x & 31;
  const int64_t j = xcd_tile(blockIdx.x, per_xcd) * 8 + (threadIdx.x >> 5);
  float gt = 0.f;
  if (j < N) {
    const int e0 = rowptr_s[j], e1 = rowptr_s[j + 1];
    const float t = tptr[0];
    const float4 xj4 = ld4(x, j, l), dj4 = ld4(dout, j, l);
    const float xj[4] = {xj4.x, xj4.y, xj4.z, xj4.w}, dj[4] = {dj4.x, dj4.y, dj4.z, dj4.w};
    float m[4], tm[4], g1[4] = {0.f, 0.f, 0.f, 0.f}, g2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) { m[q] = relu_eps(xj[q], eps); tm[q] = t * LOG2E * m[q]; }
    for (int eb = e0; eb < e1; eb += CH) {
      if (e1 - eb >= CH) bwd_chunk<CH, true>(dout, lse, agg, col_s, eb, CH, l, m, tm, g1, g2);
      else bwd_chunk<CH, false>(dout, lse, agg, col_s, eb, e1 - eb, l, m, tm, g1, g2);
    }
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r[q] = dj[q] + (xj[q] > 0.f ? fmaf(t, g2[q], g1[q]) : 0.f);
      gt = fmaf(m[q], g2[q], gt);
    }
    *reinterpret_cast<float4*>(dx + j * C + 4 * l) = make_float4(r[0], r[1], r[2], r[3]);
  }
  dt_partial_store(gt, dt_part);
}

static inline unsigned per_xcd_tiles(int64_t N, int nodes_per_wg) {
  const int64_t tiles = (N + nodes_per_wg - 1) / nodes_per_wg;
  return (unsigned)((tiles + 7) / 8);
}

extern "C" int advmil_genconv_fwd(const float* x, const int32_t* rowptr_dst, const int32_t* col_src, const float* t, float eps,
                                  int64_t N, int64_t C, float* out, float* lse, float* agg, advmil_stream_t stream) {
  if (!x || !rowptr_dst || !col_src || !t || !out || (!lse != !agg) || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  if (C == 128 && !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)lse | (uintptr_t)agg) & 15)) {
    const unsigned px = per_xcd_tiles(N, 8);
    if (lse)
      hipLaunchKernelGGL(genconv_fwd128_kernel<true>, dim3(8 * px), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src, t, eps, N,
                         px, out, lse, agg);
    else
      hipLaunchKernelGGL(genconv_fwd128_kernel<false>, dim3(8 * px), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src, t, eps, N,
                         px, out, lse, agg);
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  const unsigned px = per_xcd_tiles(N, 4);
  hipLaunchKernelGGL(genconv_fwd_kernel, dim3(8 * px), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src, t, eps, N, C, px, out,
                     lse, agg);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

static inline bool genconv_wide(int64_t C, const void* a, const void* b, const void* c, const void* d, const void* e) {
  return C == 128 && !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15);
}

extern "C" size_t advmil_genconv_bwd_workspace_bytes(int64_t N, int64_t C) {
  if (N <= 0 || C <= 0) return 0;
  return sizeof(float) * 8 * (size_t)per_xcd_tiles(N, 4);      // one dt partial per workgroup of the narrower (4 nodes) launch shape
}

extern "C" int advmil_genconv_bwd(const float* dout, const float* x, const float* agg, const float* lse,
                                  const int32_t* rowptr_src, const int32_t* col_dst, const float* t, float eps, int64_t N,
                                  int64_t C, float* dx, float* dt, void* ws, size_t ws_bytes, advmil_stream_t stream) {
  if (!dout || !x || !agg || !lse || !rowptr_src || !col_dst || !t || !dx || !dt || !ws || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_genconv_bwd_workspace_bytes(N, C)) return ADVMIL_EINVAL;
  float* part = (float*)ws;
  unsigned nwg;
  if (genconv_wide(C, x, agg, lse, dout, dx)) {
    const unsigned px = per_xcd_tiles(N, 8);
    nwg = 8 * px;
    hipLaunchKernelGGL(genconv_bwd128_kernel<GENCONV_BWD_CH>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, dout, x, agg, lse, rowptr_src, col_dst, t, eps,
                       N, px, dx, part);
  } else {
    const unsigned px = per_xcd_tiles(N, 4);
    nwg = 8 * px;
    hipLaunchKernelGGL(genconv_bwd_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, dout, x, agg, lse, rowptr_src, col_dst, t, eps, N,
                       C, px, dx, part);
  }
  ADVMIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(genconv_dt_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, (int)nwg, dt);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
