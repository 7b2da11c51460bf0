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

extern "C" int advmil_genconv_fwd(const float* x, const int32_t* rowptr_dst, const int32_t* col_src, const float* t, float eps,
                                  int64_t N, int64_t C, float* out, float* lse, float* m2, advmil_stream_t stream) {
  if (!x || !rowptr_dst || !col_src || !t || !out || !lse || !m2 || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(genconv_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rowptr_dst, col_src,
                     t, eps, N, C, out, lse, m2);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_genconv_bwd(const float* dout, const float* x, const float* out, const float* lse,
                                  const int32_t* rowptr_src, const int32_t* col_dst, const float* t, float eps, int64_t N,
                                  int64_t C, float* dx, advmil_stream_t stream) {
  if (!dout || !x || !out || !lse || !rowptr_src || !col_dst || !t || !dx || N <= 0 || C <= 0) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(genconv_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout, x, out, lse,
                     rowptr_src, col_dst, t, eps, N, C, dx);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
