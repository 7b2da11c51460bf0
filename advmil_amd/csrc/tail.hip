// The discriminator's bag-level tail (reference model/GANSurv.py:89-105 after the region level; model_utils.py:157-186) as one launch
// each way: [B <= 32, d <= 256] tensors, five tiny Linear layers, an inner product and a width-1 projection. As separate launches every
// layer costs its 5-10 us latency floor (launch + one load -> multiply -> store chain), 6 forward and up to 12 backward launches per
// pass, two passes per optimizer step. Here ONE workgroup of 16 waves walks both chains with the activations in LDS. B <= 32 rows are one
// MFMA row block: every product -- y = x W^T, dW = dpre^T x, dx = dpre W -- runs on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulate), one 32 x 32 output block per wave, the weights read ONCE from L2 in the fragment order (a first version with one thread
// per output element re-read every weight row B times through the texture path: 35 us per pass at 32 rows). Every output / gradient
// element is owned by one lane and the sums run in a fixed order: deterministic, no atomics. Dropout draws are the contraction
// epilogue's (rng_keep(key(seed, stream), row * N + col)). Layers whose widths do not fit the MFMA blocking (the width-1 input layer of
// the label embedding) take plain per-element loops.
#include "common.h"
#include "../../include/advmil_hip.h"

#define TAIL_NT 1024
#define TAIL_NW (TAIL_NT / 64)
#define TAIL_MAXB 32
#define TAIL_MAXW 256
#define TAIL_PITCH(w) ((w) + 4)                       // LDS row pitch of a [32][w] activation image (floats): b128 fragment reads conflict-free
#define TAIL_BUF (TAIL_MAXB * TAIL_PITCH(TAIL_MAXW))

struct TailArgs {
  advmil_dtail_t a;
};

__device__ __forceinline__ bool tail_mfma_ok(int K, int N) { return (K & 7) == 0 && (N & 31) == 0; }

// [B][w] global -> [32][pitch w + 4] LDS image, rows >= B zero
__device__ __forceinline__ void tail_load(float* __restrict__ dst, const float* __restrict__ src, int B, int w) {
  const int p = TAIL_PITCH(w);
  for (int i = threadIdx.x; i < TAIL_MAXB * w; i += TAIL_NT) {
    const int b = i / w, c = i - b * w;
    dst[b * p + c] = b < B ? src[(int64_t)b * w + c] : 0.f;
  }
}

// y = dropout(act(x W^T + bias)): x, y = LDS images (pitch K + 4 / N + 4); y also stored to L.y for rows < B
__device__ __forceinline__ void tail_dense_fwd(const advmil_dense_layer_t& L, int B, const float* __restrict__ in, float* __restrict__ out,
                                               const uint64_t* seed, const int64_t* __restrict__ rng_row) {
  const int K = L.K, N = L.N, pk = TAIL_PITCH(K), pn = TAIL_PITCH(N);
  const bool drop = seed && L.drop_p > 0.f;
  uint64_t key = 0;
  float inv = 1.f;
  if (drop) { key = rng_key(*seed, L.stream_id); inv = hw_rcp(1.f - L.drop_p); }
  if (tail_mfma_ok(K, N)) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hi = lane >> 5;
    for (int cb = wave; cb < N / 32; cb += TAIL_NW) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wrow = L.W + (int64_t)(cb * 32 + i) * K + hi * 4;
      const float* xrow = in + i * pk + hi * 4;
#pragma unroll 8
      for (int t = 0; t < K / 8; ++t) {
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + t * 8);
        const float4 x4 = *reinterpret_cast<const float4*>(xrow + t * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.x, w4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.y, w4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.z, w4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4.w, w4.w, acc, 0, 0, 0);
      }
      const int n = cb * 32 + i;
      const float bv = L.bias ? L.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = act_apply(L.act, acc[r] + bv);
        if (drop) v *= rng_keep(key, (uint64_t)(((rng_row && b < B) ? rng_row[b] : (int64_t)b) * N + n), L.drop_p, inv);
        if (b >= B) v = 0.f;
        out[b * pn + n] = v;
        if (b < B) L.y[(int64_t)b * N + n] = v;
      }
    }
    return;
  }
  for (int o = threadIdx.x; o < TAIL_MAXB * N; o += TAIL_NT) {
    const int b = o / N, n = o - b * N;
    float v = 0.f;
    if (b < B) {
      const float* w = L.W + (int64_t)n * K;
      const float* x = in + b * pk;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc += x[k] * w[k];
      v = act_apply(L.act, acc + (L.bias ? L.bias[n] : 0.f));
      if (drop) v *= rng_keep(key, (uint64_t)((rng_row ? rng_row[b] : (int64_t)b) * N + n), L.drop_p, inv);
      L.y[(int64_t)b * N + n] = v;
    }
    out[b * pn + n] = v;
  }
}

// dy (LDS image, pitch N + 4; gradient wrt the layer's output) -> dpre in place; dbias / dW accumulated into the arena; din (LDS image,
// pitch K + 4) when wanted. `in` = the layer's input as an LDS image (pitch K + 4, rows >= B zero).
__device__ __forceinline__ void tail_dense_bwd(const advmil_dense_layer_t& L, int B, float* __restrict__ dy, const float* __restrict__ in,
                                               float* __restrict__ din, const uint64_t* seed, const int64_t* __restrict__ rng_row) {
  const int K = L.K, N = L.N, pk = TAIL_PITCH(K), pn = TAIL_PITCH(N);
  const bool drop = seed && L.drop_p > 0.f;
  uint64_t key = 0;
  float inv = 1.f;
  if (drop) { key = rng_key(*seed, L.stream_id); inv = hw_rcp(1.f - L.drop_p); }
  for (int o = threadIdx.x; o < TAIL_MAXB * N; o += TAIL_NT) {
    const int b = o / N, n = o - b * N;
    float d = 0.f;
    if (b < B) {
      float f = 1.f, yy = L.y[(int64_t)b * N + n];
      if (drop) {
        f = rng_keep(key, (uint64_t)((rng_row ? rng_row[b] : (int64_t)b) * N + n), L.drop_p, inv);
        yy *= 1.f - L.drop_p;               // undo the 1/(1-p) on kept elements (dropped ones get f = 0 anyway)
      }
      d = dy[b * pn + n] * f * act_grad_from_out(L.act, yy);
    }
    dy[b * pn + n] = d;
  }
  __syncthreads();
  if (L.dbias)
    for (int n = threadIdx.x; n < N; n += TAIL_NT) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dy[b * pn + n];
      L.dbias[n] += s;
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, hi = lane >> 5;
  const bool mf = tail_mfma_ok(K, N) && (K & 31) == 0;
  if (L.dW) {
    if (mf) {
      // dW[n][k] += sum_b dpre[b][n] in[b][k]: one 32 x 32 block (n block, k block) per wave, the 32 rows b in 16 steps of 2
      const int nkb = K / 32, nblk = (N / 32) * nkb;
      for (int blk = wave; blk < nblk; blk += TAIL_NW) {
        const int nb = blk / nkb, kb = blk - nb * nkb;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ap = dy + hi * pn + nb * 32 + i;
        const float* bp = in + hi * pk + kb * 32 + i;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * pn], bp[2 * s * pk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          L.dW[(int64_t)n * K + kb * 32 + i] += acc[r];
        }
      }
    } else {
      for (int e = threadIdx.x; e < N * K; e += TAIL_NT) {
        const int n = e / K, k = e - n * K;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[b * pn + n] * in[b * pk + k];
        L.dW[e] += s;
      }
    }
  }
  if (din) {
    if (mf) {
      // din[b][k] = sum_n dpre[b][n] W[n][k]: one 32-column block of k per wave, n in steps of 8 (4 per lane half)
      for (int kb = wave; kb < K / 32; kb += TAIL_NW) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ap = dy + i * pn + hi * 4;
        const float* wp = L.W + (int64_t)(hi * 4) * K + kb * 32 + i;
#pragma unroll 4
        for (int t = 0; t < N / 8; ++t) {
          const float4 a4 = *reinterpret_cast<const float4*>(ap + t * 8);
          const float w0 = wp[(int64_t)(t * 8) * K], w1 = wp[(int64_t)(t * 8 + 1) * K], w2 = wp[(int64_t)(t * 8 + 2) * K],
                      w3 = wp[(int64_t)(t * 8 + 3) * K];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, w1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, w2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, w3, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int b = (r & 3) + 8 * (r >> 2) + 4 * hi;
          din[b * pk + kb * 32 + i] = b < B ? acc[r] : 0.f;
        }
      }
    } else {
      for (int o = threadIdx.x; o < TAIL_MAXB * K; o += TAIL_NT) {
        const int b = o / K, k = o - b * K;
        float s = 0.f;
        if (b < B)
          for (int n = 0; n < N; ++n) s += dy[b * pn + n] * L.W[(int64_t)n * K + k];
        din[b * pk + k] = s;
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(TAIL_NT) void dtail_fwd_kernel(TailArgs g) {
  __shared__ __attribute__((aligned(16))) float buf[3][TAIL_BUF];
  const advmil_dtail_t& a = g.a;
  const int B = a.B;
  // x chain: buf[0] -> buf[1] -> buf[0] ...; its output ends in hx
  tail_load(buf[0], a.xin, B, a.x[0].K);
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < a.nx; ++l) {
    tail_dense_fwd(a.x[l], B, buf[cur], buf[cur ^ 1], a.seed, a.rng_row);
    __syncthreads();
    cur ^= 1;
  }
  const float* const hx = buf[cur];
  float* in = buf[cur ^ 1];
  float* out = buf[2];
  tail_load(in, a.tin, B, a.y[0].K);
  __syncthreads();
  for (int l = 0; l < a.ny; ++l) {
    tail_dense_fwd(a.y[l], B, in, out, a.seed, a.rng_row);
    __syncthreads();
    float* t = in; in = out; out = t;
  }
  const float* const ht = in;
  const int d = a.y[a.ny - 1].N, pd = TAIL_PITCH(d);
  // out[b] = <u, ht> + <src, w> + bias: one wave per row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = wave; b < B; b += TAIL_NW) {
    float s = 0.f;
    for (int j = lane; j < d; j += 64) {
      const float uu = a.u ? a.u[(int64_t)b * d + j] : hx[b * pd + j];
      s += uu * ht[b * pd + j];
      if (a.prj_src == 1) s += hx[b * pd + j] * a.w_prj[j];
      else if (a.prj_src == 2) s += ht[b * pd + j] * a.w_prj[j];
    }
    s = wave_sum(s);
    if (lane == 0) a.out[b] = s + ((a.prj_src && a.b_prj) ? a.b_prj[0] : 0.f);
  }
}

__global__ __launch_bounds__(TAIL_NT) void dtail_bwd_kernel(TailArgs g) {
  __shared__ __attribute__((aligned(16))) float buf[4][TAIL_BUF];
  __shared__ float gs[TAIL_MAXB];
  const advmil_dtail_t& a = g.a;
  const int B = a.B;
  const int d = a.y[a.ny - 1].N, pd = TAIL_PITCH(d);
  const float* const hx_g = a.x[a.nx - 1].y;
  const float* const ht_g = a.y[a.ny - 1].y;
  if (threadIdx.x < TAIL_MAXB) gs[threadIdx.x] = threadIdx.x < B ? a.dout[threadIdx.x] : 0.f;
  __syncthreads();
  float* const dht = buf[0];
  float* const dhx = buf[1];
  // head: d ht = g u (+ g w), d hx = g ht ('bag') (+ g w), d u = g ht ('instance'); d w += sum_b g src, d bias += sum_b g
  for (int o = threadIdx.x; o < TAIL_MAXB * d; o += TAIL_NT) {
    const int b = o / d, j = o - b * d;
    float t = 0.f, x = 0.f;
    if (b < B) {
      const int64_t go = (int64_t)b * d + j;
      const float gb = gs[b];
      const float hxv = hx_g[go], htv = ht_g[go];
      const float uu = a.u ? a.u[go] : hxv;
      t = gb * uu;
      x = a.u ? 0.f : gb * htv;
      if (a.prj_src == 1) x += gb * a.w_prj[j];
      else if (a.prj_src == 2) t += gb * a.w_prj[j];
      if (a.u && a.du) a.du[go] = gb * htv;
    }
    dht[b * pd + j] = t;
    dhx[b * pd + j] = x;
  }
  if (a.prj_src && a.dw_prj)
    for (int j = threadIdx.x; j < d; j += TAIL_NT) {
      const float* src = a.prj_src == 1 ? hx_g : ht_g;
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += gs[b] * src[(int64_t)b * d + j];
      a.dw_prj[j] += s;
    }
  if (a.prj_src && a.db_prj && threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += gs[b];
    a.db_prj[0] += s;
  }
  __syncthreads();
  // y chain, last layer first: gradient in `dy`, the layer's input loaded into `in`, its input gradient into `din`
  {
    float* dy = dht;
    float* din = buf[2];
    float* in = buf[3];
    for (int l = a.ny - 1; l >= 0; --l) {
      const advmil_dense_layer_t& L = a.y[l];
      const bool want_in = l > 0 || a.dtin != nullptr;
      if (!want_in && !L.dW && !L.dbias) break;
      tail_load(in, l > 0 ? a.y[l - 1].y : a.tin, B, L.K);
      __syncthreads();
      tail_dense_bwd(L, B, dy, in, want_in ? din : nullptr, a.seed, a.rng_row);
      if (l == 0 && a.dtin) {
        const int pk = TAIL_PITCH(L.K);
        for (int i = threadIdx.x; i < B * L.K; i += TAIL_NT) a.dtin[i] = din[(i / L.K) * pk + (i % L.K)];
      }
      float* t = dy; dy = din; din = t;
    }
  }
  __syncthreads();
  // x chain (skipped when nothing upstream of it wants a gradient and its weights are frozen)
  {
    bool any = a.dxin != nullptr;
    for (int l = 0; l < a.nx; ++l) any = any || a.x[l].dW || a.x[l].dbias;
    if (!any) return;
    float* dy = dhx;
    float* din = buf[2];
    float* in = buf[3];
    for (int l = a.nx - 1; l >= 0; --l) {
      const advmil_dense_layer_t& L = a.x[l];
      bool below = a.dxin != nullptr;
      for (int m = 0; m < l; ++m) below = below || a.x[m].dW || a.x[m].dbias;
      if (!below && !L.dW && !L.dbias) break;
      tail_load(in, l > 0 ? a.x[l - 1].y : a.xin, B, L.K);
      __syncthreads();
      tail_dense_bwd(L, B, dy, in, below ? din : nullptr, a.seed, a.rng_row);
      if (l == 0 && a.dxin) {
        const int pk = TAIL_PITCH(L.K);
        for (int i = threadIdx.x; i < B * L.K; i += TAIL_NT) a.dxin[i] = din[(i / L.K) * pk + (i % L.K)];
      }
      float* t = dy; dy = din; din = t;
    }
  }
}

static int tail_check(const advmil_dtail_t* a, bool bwd) {
  if (!a || a->B <= 0 || a->B > TAIL_MAXB || a->nx < 1 || a->nx > ADVMIL_TAIL_MAXL || a->ny < 1 || a->ny > ADVMIL_TAIL_MAXL) return ADVMIL_EINVAL;
  if (!a->xin || !a->tin || a->prj_src < 0 || a->prj_src > 2 || (a->prj_src && !a->w_prj)) return ADVMIL_EINVAL;
  const advmil_dense_layer_t* chains[2] = {a->x, a->y};
  const int nl[2] = {a->nx, a->ny};
  for (int c = 0; c < 2; ++c)
    for (int l = 0; l < nl[c]; ++l) {
      const advmil_dense_layer_t& L = chains[c][l];
      if (!L.W || !L.y || L.K <= 0 || L.N <= 0 || L.K > TAIL_MAXW || L.N > TAIL_MAXW || L.act < 0 || L.act > 3) return ADVMIL_EINVAL;
      if (!(L.drop_p >= 0.f && L.drop_p < 1.f)) return ADVMIL_EINVAL;
      if ((L.K & 3) == 0 && (((uintptr_t)L.W) & 15)) return ADVMIL_EINVAL;
      if (l > 0 && chains[c][l - 1].N != L.K) return ADVMIL_EINVAL;
    }
  if (a->x[a->nx - 1].N != a->y[a->ny - 1].N) return ADVMIL_EINVAL;
  if (bwd ? !a->dout : !a->out) return ADVMIL_EINVAL;
  return ADVMIL_OK;
}

// the latency-shaped pair for the shipped widths (tail3.hip): 0 = launched, 1 = the call does not have its shape, < 0 = launch error
int advmil_dtail3_try(const advmil_dtail_t* a, bool bwd, hipStream_t stream);

extern "C" int advmil_dtail_fwd(const advmil_dtail_t* a, advmil_stream_t stream_) {
  const int rc = tail_check(a, false);
  if (rc) return rc;
  { const int r3 = advmil_dtail3_try(a, false, (hipStream_t)stream_); if (r3 <= 0) return r3 == 0 ? ADVMIL_OK : -(r3 + 1000); }
  TailArgs g;
  g.a = *a;
  hipLaunchKernelGGL(dtail_fwd_kernel, dim3(1), dim3(TAIL_NT), 0, (hipStream_t)stream_, g);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_dtail_bwd(const advmil_dtail_t* a, advmil_stream_t stream_) {
  const int rc = tail_check(a, true);
  if (rc) return rc;
  { const int r3 = advmil_dtail3_try(a, true, (hipStream_t)stream_); if (r3 <= 0) return r3 == 0 ? ADVMIL_OK : -(r3 + 1000); }
  TailArgs g;
  g.a = *a;
  hipLaunchKernelGGL(dtail_bwd_kernel, dim3(1), dim3(TAIL_NT), 0, (hipStream_t)stream_, g);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
