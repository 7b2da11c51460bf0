// K6: GENConv softmax aggregation of PatchGCN (model/backbone.py:139,157; arithmetic of torch_geometric.nn.GENConv,
// restated from its published semantics -- parity unpinned, see oracle/advmil_oracle.py::genconv):
//   message  m_j   = relu(x_j) + eps                                  (per source node j, per channel c)
//   weights  w_ij  = softmax over the in-edges j->i of (t * m_j)      (per target i, per channel c)
//   output   out_i = sum_j w_ij m_j + x_i
// HBM-bound sparse gather: one wave per node, lanes own channels (256 B coalesced row segments), neighbours come from a
// CSR list, so every x row read is a full-line read and there are no atomics: the forward walks the graph by DESTINATION,
// the backward by SOURCE (both CSR images are built once per graph on the host side).
#include "common.h"
#include "../../include/advmil_hip.h"

__global__ __launch_bounds__(256) void genconv_fwd_kernel(const float* __restrict__ x, const int* __restrict__ rowptr,
                                                          const int* __restrict__ col, const float* __restrict__ tptr,
                                                          float eps, int64_t N, int64_t C, float* __restrict__ out,
                                                          float* __restrict__ lse, float* __restrict__ m2) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float t = tptr[0];
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int64_t c = lane; c < C; c += 64) {
    float mx = -INFINITY;
    for (int e = e0; e < e1; ++e) {
      const float v = x[(int64_t)col[e] * C + c];
      mx = fmaxf(mx, t * ((v > 0.f ? v : 0.f) + eps));
    }
    float den = 0.f, a1 = 0.f, a2 = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float v = x[(int64_t)col[e] * C + c];
      const float m = (v > 0.f ? v : 0.f) + eps;
      const float w = hw_exp(t * m - mx);
      den += w; a1 += w * m; a2 += w * m * m;
    }
    const bool has = e1 > e0;
    const float iden = has ? hw_rcp(den) : 0.f;
    const float agg = a1 * iden;
    out[i * C + c] = agg + x[i * C + c];
    lse[i * C + c] = has ? mx + hw_log(den) : 0.f;
    m2[i * C + c] = a2 * iden;
  }
}

// dx_j = dout_j + relu'(x_j) * sum_{edges j->i} dout_i * w_ij * (1 + t (m_j - agg_i)),  agg_i = out_i - x_i
__global__ __launch_bounds__(256) void genconv_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                          const float* __restrict__ out, const float* __restrict__ lse,
                                                          const int* __restrict__ rowptr_s, const int* __restrict__ col_s,
                                                          const float* __restrict__ tptr, float eps, int64_t N, int64_t C,
                                                          float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  const float t = tptr[0];
  const int e0 = rowptr_s[j], e1 = rowptr_s[j + 1];
  for (int64_t c = lane; c < C; c += 64) {
    const float xj = x[j * C + c];
    const float m = (xj > 0.f ? xj : 0.f) + eps;
    float g = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int64_t i = col_s[e];
      const float agg = out[i * C + c] - x[i * C + c];
      const float w = hw_exp(t * m - lse[i * C + c]);
      g += dout[i * C + c] * w * (1.f + t * (m - agg));
    }
    dx[j * C + c] = dout[j * C + c] + (xj > 0.f ? g : 0.f);
  }
}

// ---- C == 128 (the PatchGCN width, model/backbone.py:40): a 32-lane half-wave owns a node, one float4 (16 B) per lane covers its
// 512-byte row, 8 nodes per workgroup; the neighbour rows are gathered ONCE (up to 8 in flight, the k-NN degree of
// tools/patchgcn_graph_s2.py; longer lists are merged chunk by chunk with the online-softmax rescale).
__device__ __forceinline__ float4 f4_msg(const float4 v, float eps) {
  return make_float4((v.x > 0.f ? v.x : 0.f) + eps, (v.y > 0.f ? v.y : 0.f) + eps, (v.z > 0.f ? v.z : 0.f) + eps, (v.w > 0.f ? v.w : 0.f) + eps);
}
__global__ __launch_bounds__(256) void genconv_fwd128_kernel(const float* __restrict__ x, const int* __restrict__ rowptr,
                                                             const int* __restrict__ col, const float* __restrict__ tptr, float eps,
                                                             int64_t N, float* __restrict__ out, float* __restrict__ lse,
                                                             float* __restrict__ m2) {
  constexpr int C = 128;
  const int l = threadIdx.x & 31;
  const int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (i >= N) return;
  const float t = tptr[0];
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f},
        a2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int eb = e0; eb < e1; eb += 8) {
    float4 m[8];
    const int n = (e1 - eb) < 8 ? (e1 - eb) : 8;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      m[k] = k < n ? f4_msg(*reinterpret_cast<const float4*>(x + (int64_t)col[eb + k] * C + 4 * l), eps) : make_float4(0.f, 0.f, 0.f, 0.f);
    float cm[4] = {mx[0], mx[1], mx[2], mx[3]};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < n) {
        cm[0] = fmaxf(cm[0], t * m[k].x); cm[1] = fmaxf(cm[1], t * m[k].y);
        cm[2] = fmaxf(cm[2], t * m[k].z); cm[3] = fmaxf(cm[3], t * m[k].w);
      }
    if (eb > e0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float sc = hw_exp(mx[q] - cm[q]);
        den[q] *= sc; a1[q] *= sc; a2[q] *= sc;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) mx[q] = cm[q];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < n) {
        const float mv[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float w = hw_exp(t * mv[q] - mx[q]);
          den[q] += w; a1[q] += w * mv[q]; a2[q] += w * mv[q] * mv[q];
        }
      }
  }
  const bool has = e1 > e0;
  const float4 xi = *reinterpret_cast<const float4*>(x + i * C + 4 * l);
  const float xv[4] = {xi.x, xi.y, xi.z, xi.w};
  float o[4], ls[4], mm[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float iden = has ? hw_rcp(den[q]) : 0.f;
    o[q] = a1[q] * iden + xv[q];
    ls[q] = has ? mx[q] + hw_log(den[q]) : 0.f;
    mm[q] = a2[q] * iden;
  }
  *reinterpret_cast<float4*>(out + i * C + 4 * l) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(lse + i * C + 4 * l) = make_float4(ls[0], ls[1], ls[2], ls[3]);
  *reinterpret_cast<float4*>(m2 + i * C + 4 * l) = make_float4(mm[0], mm[1], mm[2], mm[3]);
}

__global__ __launch_bounds__(256) void genconv_bwd128_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                             const float* __restrict__ out, const float* __restrict__ lse,
                                                             const int* __restrict__ rowptr_s, const int* __restrict__ col_s,
                                                             const float* __restrict__ tptr, float eps, int64_t N,
                                                             float* __restrict__ dx) {
  constexpr int C = 128;
  const int l = threadIdx.x & 31;
  const int64_t j = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (j >= N) return;
  const float t = tptr[0];
  const int e0 = rowptr_s[j], e1 = rowptr_s[j + 1];
  const float4 xj4 = *reinterpret_cast<const float4*>(x + j * C + 4 * l);
  const float xj[4] = {xj4.x, xj4.y, xj4.z, xj4.w};
  float m[4], g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) m[q] = (xj[q] > 0.f ? xj[q] : 0.f) + eps;
  for (int e = e0; e < e1; ++e) {
    const int64_t i = col_s[e];
    const float4 o4 = *reinterpret_cast<const float4*>(out + i * C + 4 * l), x4 = *reinterpret_cast<const float4*>(x + i * C + 4 * l);
    const float4 l4 = *reinterpret_cast<const float4*>(lse + i * C + 4 * l), d4 = *reinterpret_cast<const float4*>(dout + i * C + 4 * l);
    const float ov[4] = {o4.x, o4.y, o4.z, o4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w}, lv[4] = {l4.x, l4.y, l4.z, l4.w},
                dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float agg = ov[q] - xv[q];
      const float w = hw_exp(t * m[q] - lv[q]);
      g[q] += dv[q] * w * (1.f + t * (m[q] - agg));
    }
  }
  const float4 dj = *reinterpret_cast<const float4*>(dout + j * C + 4 * l);
  *reinterpret_cast<float4*>(dx + j * C + 4 * l) = make_float4(dj.x + (xj[0] > 0.f ? g[0] : 0.f), dj.y + (xj[1] > 0.f ? g[1] : 0.f),
                                                                dj.z + (xj[2] > 0.f ? g[2] : 0.f), dj.w + (xj[3] > 0.f ? g[3] : 0.f));
}

extern "C" int advmil_genconv_fwd(const float* x, const int32_t* rowptr_dst, const int32_t* col_src, const float* t, float eps,
                                  int64_t N, int64_t C, float* out, float* lse, float* m2, advmil_stream_t stream) {
  if (!x || !rowptr_dst || !col_src || !t || !out || !lse || !m2 || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  if (C == 128 && !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)lse | (uintptr_t)m2) & 15)) {
    hipLaunchKernelGGL(genconv_fwd128_kernel, dim3((unsigned)((N + 7) / 8)), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src, t,
                       eps, N, out, lse, m2);
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  hipLaunchKernelGGL(genconv_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src,
                     t, eps, N, C, out, lse, m2);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_genconv_bwd(const float* dout, const float* x, const float* out, const float* lse,
                                  const int32_t* rowptr_src, const int32_t* col_dst, const float* t, float eps, int64_t N,
                                  int64_t C, float* dx, advmil_stream_t stream) {
  if (!dout || !x || !out || !lse || !rowptr_src || !col_dst || !t || !dx || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  if (C == 128 && !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)lse | (uintptr_t)dout | (uintptr_t)dx) & 15)) {
    hipLaunchKernelGGL(genconv_bwd128_kernel, dim3((unsigned)((N + 7) / 8)), dim3(256), 0, (hipStream_t)stream, dout, x, out, lse,
                       rowptr_src, col_dst, t, eps, N, dx);
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  hipLaunchKernelGGL(genconv_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout, x, out, lse,
                     rowptr_src, col_dst, t, eps, N, C, dx);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
