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

// Walk: a workgroup takes a tile of 8 S consecutive nodes (S <= 16, chosen by the launcher so that the grid still covers the chip twice);
// their row pointers and their (contiguous) edge list are read once, coalesced, into LDS. Each half-wave then walks its S nodes as a flat
// sequence of edge chunks, software-pipelined: the rows of chunk c+1 are requested before the arithmetic of chunk c starts (two register
// buffers), so a wave always has a full chunk of gathers in flight and no walk ever waits on a rowptr -> col -> row chain. (One chunk at a
// time per wave, as first written, measured 41 / 66 us however the arithmetic or the written bytes were changed: launch time = rounds of
// workgroups x three dependent memory latencies.)
constexpr int GT_MAXS = 16;                 // nodes per half-wave, at most
constexpr int GT_EDGES = 2048;              // edge indices staged per tile (16 per node at S = 16; the 8-NN graphs need 1024)

struct TileEdges {
  int rp[8 * GT_MAXS + 1];
  int col[GT_EDGES];
  // stage the tile's row pointers and the head of its edge list (all 256 threads; nt = nodes of the tile)
  __device__ __forceinline__ void stage(const int* __restrict__ rowptr, const int* __restrict__ colg, int64_t i0, int nt) {
    if ((int)threadIdx.x <= nt) rp[threadIdx.x] = rowptr[i0 + threadIdx.x];
    __syncthreads();
    const int ebase = rp[0];
    const int ne = (rp[nt] - ebase) < GT_EDGES ? (rp[nt] - ebase) : GT_EDGES;
    for (int k = threadIdx.x; k < ne; k += 256) col[k] = colg[ebase + k];
    __syncthreads();
  }
  // INLDS: the whole edge list of the tile was staged (decided per workgroup); otherwise every index comes from global memory
  // (the LDS read is unconditional and clamped -- a predicated read costs a branch and its own lgkmcnt wait per edge; the caller selects)
  template <bool INLDS>
  __device__ __forceinline__ int64_t row_of(const int* __restrict__ colg, int e, bool valid, int64_t otherwise) const {
    if (INLDS) {
      int k = e - rp[0];
      k = k < GT_EDGES - 1 ? k : GT_EDGES - 1;
      const int v = col[k];
      return valid ? (int64_t)v : otherwise;
    }
    return valid ? (int64_t)colg[e] : otherwise;
  }
  __device__ __forceinline__ bool staged(int nt) const { return rp[nt] - rp[0] <= GT_EDGES; }
};

// a chunk of <= CH edges of node r of the tile: [eb, eb + n) of the CSR list; first / last chunk of its node
struct Chunk {
  int r, eb, n;
  bool first, last;
};
template <int CH>
__device__ __forceinline__ Chunk node_chunk(const TileEdges& te, int r, int nt) {
  Chunk c;
  c.r = r < nt ? r : -1;
  const int rr = r < nt ? r : 0;
  const int e0 = te.rp[rr], e1 = te.rp[rr + 1];
  c.eb = e0; c.n = (e1 - e0) < CH ? (e1 - e0) : CH;
  c.first = true; c.last = e0 + CH >= e1;
  return c;
}
template <int CH>
__device__ __forceinline__ Chunk next_chunk(const TileEdges& te, const Chunk& c, int nt) {
  if (c.last) return node_chunk<CH>(te, c.r + 8, nt);
  Chunk d;
  const int e1 = te.rp[c.r + 1];
  d.r = c.r; d.eb = c.eb + CH; d.n = (e1 - d.eb) < CH ? (e1 - d.eb) : CH;
  d.first = false; d.last = d.eb + CH >= e1;
  return d;
}

// ---- forward
struct FwdRows {
  float4 v[8], own;
};
struct FwdState {
  float mm[4], mx[4], den[4], a1[4], xi[4];
};
template <bool INLDS>
__device__ __forceinline__ void fwd_load(const TileEdges& te, const float* __restrict__ x, const int* __restrict__ col, int64_t i0,
                                         const Chunk& c, int l, FwdRows& b) {
  // the same number of loads whatever the chunk (the compiler can then wait on the OLDER buffer with a constant vmcnt while these are in
  // flight): past the list, past the tile's last chunk and for the own row of a non-first chunk the address is the node's own row (an L1 hit)
  const int64_t self = i0 + (c.r < 0 ? 0 : c.r);
  const int n = c.r < 0 ? 0 : c.n;
#pragma unroll
  for (int k = 0; k < 8; ++k) b.v[k] = ld4(x, te.row_of<INLDS>(col, c.eb + k, k < n, self), l);
  b.own = ld4(x, self, l);
}
template <bool FULL, bool POS>
__device__ __forceinline__ void fwd_math(const FwdRows& b, int n, bool first, float eps, float tl, FwdState& st) {
  float m[8][4];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    m[k][0] = relu_eps(b.v[k].x, eps); m[k][1] = relu_eps(b.v[k].y, eps); m[k][2] = relu_eps(b.v[k].z, eps); m[k][3] = relu_eps(b.v[k].w, eps);
  }
  if (FULL) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) st.mm[q] = POS ? max3f(st.mm[q], m[k][q], m[k + 1][q]) : min3f(st.mm[q], m[k][q], m[k + 1][q]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st.mm[q] = POS ? max3f(st.mm[q], m[k][q], m[k][q]) : min3f(st.mm[q], m[k][q], m[k][q]);
      }
  }
  float cm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cm[q] = tl * st.mm[q];
  if (!first) {
    float e[4], sc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = st.mx[q] - cm[q];
    exp2x4(e, sc);
#pragma unroll
    for (int q = 0; q < 4; ++q) { st.den[q] *= sc[q]; st.a1[q] *= sc[q]; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) st.mx[q] = cm[q];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (FULL || k < n) {
      float e[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = fmaf(m[k][q], tl, -st.mx[q]);
      exp2x4(e, w);
#pragma unroll
      for (int q = 0; q < 4; ++q) { st.den[q] += w[q]; st.a1[q] = fmaf(w[q], m[k][q], st.a1[q]); }
    }
  }
}
template <bool SAVE>
__device__ __forceinline__ void fwd_step(const FwdRows& b, const Chunk& c, int64_t i0, int l, float eps, float tl, FwdState& st,
                                         float* __restrict__ out, float* __restrict__ lse, float* __restrict__ agg_out) {
  const bool pos = tl >= 0.f;
  if (c.first) {
    st.xi[0] = b.own.x; st.xi[1] = b.own.y; st.xi[2] = b.own.z; st.xi[3] = b.own.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) { st.mm[q] = pos ? -INFINITY : INFINITY; st.mx[q] = 0.f; st.den[q] = 0.f; st.a1[q] = 0.f; }
  }
  if (c.n == 8 && pos) fwd_math<true, true>(b, 8, c.first, eps, tl, st);
  else if (pos) { if (c.n > 0) fwd_math<false, true>(b, c.n, c.first, eps, tl, st); }
  else if (c.n > 0) fwd_math<false, false>(b, c.n, c.first, eps, tl, st);
  if (c.last) {
    const bool has = c.n > 0 || !c.first;
    float o[4], ls[4], ag[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ag[q] = has ? st.a1[q] * hw_rcp(st.den[q]) : 0.f;
      o[q] = ag[q] + st.xi[q];
      ls[q] = has ? st.mx[q] + hw_log2(st.den[q]) : 0.f;
    }
    const int64_t off = (i0 + c.r) * 128 + 4 * l;
    *reinterpret_cast<float4*>(out + off) = make_float4(o[0], o[1], o[2], o[3]);
    if (SAVE) {
      *reinterpret_cast<float4*>(lse + off) = make_float4(ls[0], ls[1], ls[2], ls[3]);
      *reinterpret_cast<float4*>(agg_out + off) = make_float4(ag[0], ag[1], ag[2], ag[3]);
    }
  }
}

template <bool SAVE, bool INLDS>
__device__ __forceinline__ void fwd_walk(const TileEdges& te, const float* __restrict__ x, const int* __restrict__ col, int64_t i0, int nt,
                                         int h, int l, float eps, float tl, float* __restrict__ out, float* __restrict__ lse,
                                         float* __restrict__ agg_out) {
  FwdRows A, B;
  FwdState st;
  Chunk ca = node_chunk<8>(te, h, nt), cb;
  if (ca.r < 0) return;
  fwd_load<INLDS>(te, x, col, i0, ca, l, A);
  while (true) {
    cb = next_chunk<8>(te, ca, nt);
    fwd_load<INLDS>(te, x, col, i0, cb, l, B);
    fwd_step<SAVE>(A, ca, i0, l, eps, tl, st, out, lse, agg_out);
    if (cb.r < 0) break;
    ca = next_chunk<8>(te, cb, nt);
    fwd_load<INLDS>(te, x, col, i0, ca, l, A);
    fwd_step<SAVE>(B, cb, i0, l, eps, tl, st, out, lse, agg_out);
    if (ca.r < 0) break;
  }
}

template <bool SAVE>
__global__ __launch_bounds__(256) void genconv_fwd128_kernel(const float* __restrict__ x, const int* __restrict__ rowptr,
                                                             const int* __restrict__ col, const float* __restrict__ tptr, float eps,
                                                             int64_t N, int tile_nodes, unsigned per_xcd, float* __restrict__ out,
                                                             float* __restrict__ lse, float* __restrict__ agg_out) {
  __shared__ TileEdges te;
  const int l = threadIdx.x & 31, h = threadIdx.x >> 5;
  const int64_t i0 = xcd_tile(blockIdx.x, per_xcd) * tile_nodes;
  if (i0 >= N) return;
  const int nt = (N - i0) < tile_nodes ? (int)(N - i0) : tile_nodes;
  te.stage(rowptr, col, i0, nt);
  const float tl = tptr[0] * LOG2E;
  if (te.staged(nt)) fwd_walk<SAVE, true>(te, x, col, i0, nt, h, l, eps, tl, out, lse, agg_out);
  else fwd_walk<SAVE, false>(te, x, col, i0, nt, h, l, eps, tl, out, lse, agg_out);
}

// ---- backward, per source node j and channel:  g1 = sum_i dw_i,  g2 = sum_i dw_i (m_j - agg_i),  dw_i = dout_i exp2(tl m_j - lse2_i)
//   dx_j = dout_j + relu'(x_j) (g1 + t g2),   dt += m_j g2
template <int CH>
struct BwdRows {
  float4 d[CH], ls[CH], ag[CH], xj, dj;
};
struct BwdState {
  float m[4], tm[4], g1[4], g2[4], xj[4], dj[4];
};
template <int CH, bool INLDS>
__device__ __forceinline__ void bwd_load(const TileEdges& te, const float* __restrict__ dout, const float* __restrict__ x,
                                         const float* __restrict__ agg, const float* __restrict__ lse, const int* __restrict__ col_s,
                                         int64_t j0, const Chunk& c, int l, BwdRows<CH>& b) {
  const int64_t self = j0 + (c.r < 0 ? 0 : c.r);       // (constant load count per chunk: see fwd_load)
  const int n = c.r < 0 ? 0 : c.n;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int64_t i = te.row_of<INLDS>(col_s, c.eb + k, k < n, self);
    b.d[k] = ld4(dout, i, l); b.ls[k] = ld4(lse, i, l); b.ag[k] = ld4(agg, i, l);
  }
  b.xj = ld4(x, self, l); b.dj = ld4(dout, self, l);
}
template <int CH, bool FULL>
__device__ __forceinline__ void bwd_math(const BwdRows<CH>& b, int n, BwdState& st) {
#pragma unroll
  for (int k = 0; k < CH; ++k)
    if (FULL || k < n) {
      const float dv[4] = {b.d[k].x, b.d[k].y, b.d[k].z, b.d[k].w}, lv[4] = {b.ls[k].x, b.ls[k].y, b.ls[k].z, b.ls[k].w},
                  av[4] = {b.ag[k].x, b.ag[k].y, b.ag[k].z, b.ag[k].w};
      float e[4], w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = st.tm[q] - lv[q];
      exp2x4(e, w);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dw = dv[q] * w[q];
        st.g1[q] += dw;
        st.g2[q] = fmaf(dw, st.m[q] - av[q], st.g2[q]);
      }
    }
}
template <int CH>
__device__ __forceinline__ void bwd_step(const BwdRows<CH>& b, const Chunk& c, int64_t j0, int l, float eps, float t, BwdState& st,
                                         float* __restrict__ dx, float& gt) {
  if (c.first) {
    st.xj[0] = b.xj.x; st.xj[1] = b.xj.y; st.xj[2] = b.xj.z; st.xj[3] = b.xj.w;
    st.dj[0] = b.dj.x; st.dj[1] = b.dj.y; st.dj[2] = b.dj.z; st.dj[3] = b.dj.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) { st.m[q] = relu_eps(st.xj[q], eps); st.tm[q] = t * LOG2E * st.m[q]; st.g1[q] = 0.f; st.g2[q] = 0.f; }
  }
  if (c.n == CH) bwd_math<CH, true>(b, CH, st);
  else if (c.n > 0) bwd_math<CH, false>(b, c.n, st);
  if (c.last) {
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r[q] = st.dj[q] + (st.xj[q] > 0.f ? fmaf(t, st.g2[q], st.g1[q]) : 0.f);
      gt = fmaf(st.m[q], st.g2[q], gt);
    }
    *reinterpret_cast<float4*>(dx + (j0 + c.r) * 128 + 4 * l) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

template <int CH, bool INLDS>
__device__ __forceinline__ void bwd_walk(const TileEdges& te, const float* __restrict__ dout, const float* __restrict__ x,
                                         const float* __restrict__ agg, const float* __restrict__ lse, const int* __restrict__ col_s,
                                         int64_t j0, int nt, int h, int l, float eps, float t, float* __restrict__ dx, float& gt) {
  BwdRows<CH> A, B;
  BwdState st;
  Chunk ca = node_chunk<CH>(te, h, nt), cb;
  if (ca.r < 0) return;
  bwd_load<CH, INLDS>(te, dout, x, agg, lse, col_s, j0, ca, l, A);
  while (true) {
    cb = next_chunk<CH>(te, ca, nt);
    bwd_load<CH, INLDS>(te, dout, x, agg, lse, col_s, j0, cb, l, B);
    bwd_step<CH>(A, ca, j0, l, eps, t, st, dx, gt);
    if (cb.r < 0) break;
    ca = next_chunk<CH>(te, cb, nt);
    bwd_load<CH, INLDS>(te, dout, x, agg, lse, col_s, j0, ca, l, A);
    bwd_step<CH>(B, cb, j0, l, eps, t, st, dx, gt);
    if (ca.r < 0) break;
  }
}

template <int CH>
__global__ __launch_bounds__(256) void genconv_bwd128_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                             const float* __restrict__ agg, const float* __restrict__ lse,
                                                             const int* __restrict__ rowptr_s, const int* __restrict__ col_s,
                                                             const float* __restrict__ tptr, float eps, int64_t N, int tile_nodes,
                                                             unsigned per_xcd, float* __restrict__ dx, float* __restrict__ dt_part) {
  __shared__ TileEdges te;
  const int l = threadIdx.x & 31, h = threadIdx.x >> 5;
  const int64_t j0 = xcd_tile(blockIdx.x, per_xcd) * tile_nodes;
  float gt = 0.f;
  if (j0 < N) {                                        // (uniform over the workgroup: the barriers inside stage() are safe)
    const int nt = (N - j0) < tile_nodes ? (int)(N - j0) : tile_nodes;
    te.stage(rowptr_s, col_s, j0, nt);
    const float t = tptr[0];
    if (te.staged(nt)) bwd_walk<CH, true>(te, dout, x, agg, lse, col_s, j0, nt, h, l, eps, t, dx, gt);
    else bwd_walk<CH, false>(te, dout, x, agg, lse, col_s, j0, nt, h, l, eps, t, dx, gt);
  }
  dt_partial_store(gt, dt_part);
}

// nodes per workgroup tile of the C == 128 kernels: 8 S, the largest S <= 16 that still leaves two workgroups per CU (512 tiles)
static inline int tile_nodes_for(int64_t N) {
  int64_t s = N / (8 * 512);
  s = s < 1 ? 1 : (s > GT_MAXS ? GT_MAXS : s);
  return (int)(8 * s);
}
static inline unsigned per_xcd_tiles(int64_t N, int nodes_per_wg) {
  const int64_t tiles = (N + nodes_per_wg - 1) / nodes_per_wg;
  return (unsigned)((tiles + 7) / 8);
}

extern "C" int advmil_genconv_fwd(const float* x, const int32_t* rowptr_dst, const int32_t* col_src, const float* t, float eps,
                                  int64_t N, int64_t C, float* out, float* lse, float* agg, advmil_stream_t stream) {
  if (!x || !rowptr_dst || !col_src || !t || !out || (!lse != !agg) || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  if (C == 128 && !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)lse | (uintptr_t)agg) & 15)) {
    const int tn = tile_nodes_for(N);
    const unsigned px = per_xcd_tiles(N, tn);
    if (lse)
      hipLaunchKernelGGL(genconv_fwd128_kernel<true>, dim3(8 * px), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src, t, eps, N,
                         tn, px, out, lse, agg);
    else
      hipLaunchKernelGGL(genconv_fwd128_kernel<false>, dim3(8 * px), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src, t, eps, N,
                         tn, px, out, lse, agg);
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
    const int tn = tile_nodes_for(N);
    const unsigned px = per_xcd_tiles(N, tn);
    nwg = 8 * px;
    hipLaunchKernelGGL(genconv_bwd128_kernel<GENCONV_BWD_CH>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, dout, x, agg, lse, rowptr_src,
                       col_dst, t, eps, N, tn, px, dx, part);
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
