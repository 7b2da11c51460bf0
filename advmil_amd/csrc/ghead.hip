// The generator's bag-level head as two launches each way (reference model/GANSurv.py:13-46 Generator.forward after the backbone's pooling:
// ABMIL's rho = Linear(d0, d1) -> ReLU -> Dropout (model/backbone.py:66-70), MLPs[0] = Linear(., d2) -> ReLU -> Dropout, noise of width d2
// concatenated, MLPs[1] = Linear(2 d2, 1), out_scale sigmoid; model/model_utils.py:124-140 make_noise_mlp_layer with hops = 1, noise = [0, 1]).
// Layer by layer this is 5 launches forward and 9 backward on [B <= 32, d] tensors -- 43 + 82 us of a 3.0 ms step at 16 bags, 23 + 43 us of a
// 0.68 ms step at one bag -- each a launch + load -> use -> store latency chain. The weights (885 KB) are too many for one workgroup's load
// path, so the work is cut along the FIRST hidden layer's output columns instead of along layers:
//   forward  A (ds / 16 workgroups): workgroup j owns 16 columns of the first hidden layer ("slice layer": rho, or MLPs[0] when the backbone
//              has no rho): hs[:, cj] = dropout(relu(X Ws[cj]^T + bs[cj])), and -- the next layer being linear in hs -- that slice's share
//              of the next layer's pre-activation (K-split): P_j[B, d2] = hs[:, cj] W0[:, cj]^T (or the slice's share of the output dot);
//            B (one workgroup per bag): sums the shares in a fixed order, bias, ReLU, dropout, the output dot with the noise half drawn in
//              the kernel (the draws of advmil_uniform_fill at the same site), out_scale.
//   backward C (ds / 16 workgroups): every workgroup recomputes the [B, d2] gradient at the second layer (cheap), then owns its 16 columns:
//              dhs, the slice's rows / columns of both weight gradients and the bias gradient ADDED in place into the arena (no partials:
//              a column belongs to one workgroup), and its share of dX (K-split again); workgroup 0 also does the output layer's gradients;
//            D (one workgroup per bag): sums the dX shares.
// fp32 FMA arithmetic, every sum in a fixed order. B <= 32, d0 <= 512, ds % 16 == 0, d2 <= 256.
#include <cstdlib>
#include "common.h"
#include "../../include/advmil_hip.h"

#define GH_NT 256
#define GH_CW 16

#define LDS_BARRIER()                                   \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
  } while (0)

struct GHeadArgs {
  advmil_ghead_t a;
  int nw;          // slice workgroups = ds / 16
};

__device__ __forceinline__ float gh_dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// ------------------------------------------------------------------------------------------------------------------------------------
// forward A
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GH_NT) void ghead_fwd_slices_kernel(GHeadArgs g) {
  extern __shared__ float4 gh_smem4[];
  float* const sm = reinterpret_cast<float*>(gh_smem4);
  const advmil_ghead_t& a = g.a;
  const int B = a.B, d0 = a.d0, d2 = a.d2;
  const bool two = a.d1 > 0;
  const int ds = two ? a.d1 : a.d2;
  const float* const Ws = two ? a.Wr : a.W0;
  const float* const bs = two ? a.br : a.b0;
  const float ps = two ? a.p1 : a.p2;
  const uint64_t sids = two ? a.sid1 : a.sid2;
  const int tid = threadIdx.x, j = blockIdx.x, c0 = j * GH_CW;
  const int PX = d0 + 4, q0 = d0 >> 2;
  float* const sX = sm;                       // [B][PX]
  float* const sW = sX + B * PX;              // [16][PX]
  float* const sH = sW + GH_CW * PX;          // [32][16]
  // ---- global reads, all up front
  float4 w0[4];
  if (two && tid < d2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w0[q] = *reinterpret_cast<const float4*>(a.W0 + (int64_t)tid * ds + c0 + 4 * q);
  }
  for (int o = tid; o < B * q0; o += GH_NT) {
    const int b = o / q0, c = o - b * q0;
    *reinterpret_cast<float4*>(sX + b * PX + 4 * c) = *reinterpret_cast<const float4*>(a.x + (int64_t)b * a.ldx + 4 * c);
  }
  for (int o = tid; o < GH_CW * q0; o += GH_NT) {
    const int r = o / q0, c = o - r * q0;
    *reinterpret_cast<float4*>(sW + r * PX + 4 * c) = *reinterpret_cast<const float4*>(Ws + (int64_t)(c0 + r) * d0 + 4 * c);
  }
  const int c = tid & 15, bq = tid >> 4;
  const float bias = bs ? bs[c0 + c] : 0.f;
  const bool drop = a.seed && ps > 0.f;
  const uint64_t key = drop ? rng_key(*a.seed, sids) : 0;
  const float inv = drop ? hw_rcp(1.0f - ps) : 1.f;
  LDS_BARRIER();
  // ---- the slice layer: thread (c, bq) -> rows bq and bq + 16
  for (int b = bq; b < B; b += 16) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xr = sX + b * PX;
    const float* wr = sW + c * PX;
    for (int k = 0; k < q0; ++k) {
      const float4 x = *reinterpret_cast<const float4*>(xr + 4 * k), w = *reinterpret_cast<const float4*>(wr + 4 * k);
      acc.x += x.x * w.x; acc.y += x.y * w.y; acc.z += x.z * w.z; acc.w += x.w * w.w;
    }
    float v = fmaxf((acc.x + acc.y) + (acc.z + acc.w) + bias, 0.f);
    if (drop) v *= rng_keep(key, (uint64_t)((a.rng_row ? a.rng_row[b] : (int64_t)b) * ds + c0 + c), ps, inv);
    sH[b * GH_CW + c] = v;
    a.hs[(int64_t)b * ds + c0 + c] = v;
  }
  LDS_BARRIER();
  // ---- this slice's share of the next layer
  if (two) {
    if (tid < d2) {
      for (int b = 0; b < B; ++b) {
        const float4* h = reinterpret_cast<const float4*>(sH + b * GH_CW);
        const float p = (gh_dot4(h[0], w0[0]) + gh_dot4(h[1], w0[1])) + (gh_dot4(h[2], w0[2]) + gh_dot4(h[3], w0[3]));
        a.ws[((int64_t)j * B + b) * d2 + tid] = p;
      }
    }
  } else if (tid < B) {
    float z = 0.f;
#pragma unroll
    for (int cc = 0; cc < GH_CW; ++cc) z += sH[tid * GH_CW + cc] * a.W1[c0 + cc];
    a.ws[(int64_t)j * B + tid] = z;
  }
}

// value of the noise input (b, n): 0 (modes 0, 1), the caller's tensor (2), the counter RNG's uniform draw (3)
__device__ __forceinline__ float gh_noise(const advmil_ghead_t& a, uint64_t keyn, int b, int n) {
  if (a.noise_mode == 2) return a.noise[(int64_t)b * a.d2 + n];
  if (a.noise_mode == 3) return rng_uniform(keyn, (uint64_t)((a.rng_row ? a.rng_row[b] : (int64_t)b) * a.d2 + n));
  return 0.f;
}

__device__ __forceinline__ float gh_block_sum(float t, float* red) {      // 256 threads, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  LDS_BARRIER();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// forward B: one workgroup per bag
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GH_NT) void ghead_fwd_finish_kernel(GHeadArgs g) {
  __shared__ float red[4];
  const advmil_ghead_t& a = g.a;
  const int B = a.B, d2 = a.d2, b = blockIdx.x, n = threadIdx.x, nw = g.nw;
  const bool two = a.d1 > 0;
  float t = 0.f;
  if (two) {
    if (n < d2) {
      float s = a.b0 ? a.b0[n] : 0.f;
      for (int j = 0; j < nw; j += 8) {          // eight shares in flight (a share-by-share loop is one memory round trip per share)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = j + u < nw ? a.ws[((int64_t)(j + u) * B + b) * d2 + n] : 0.f;
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      s = fmaxf(s, 0.f);
      if (a.seed && a.p2 > 0.f)
        s *= rng_keep(rng_key(*a.seed, a.sid2), (uint64_t)((a.rng_row ? a.rng_row[b] : (int64_t)b) * d2 + n), a.p2, hw_rcp(1.0f - a.p2));
      a.h2[(int64_t)b * d2 + n] = s;
      t = s * a.W1[n];
    }
  } else if (n < nw) {
    t = a.ws[(int64_t)n * B + b];
  }
  if (a.noise_mode >= 2 && n < d2) t += gh_noise(a, a.noise_mode == 3 ? rng_key(*a.seed, a.sid_noise) : 0, b, n) * a.W1[d2 + n];
  const float z = gh_block_sum(t, red) + (a.b1 ? a.b1[0] : 0.f);
  if (n == 0) a.pred[b] = a.out_act ? act_apply(ACT_SIGMOID, z) : z;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// backward C
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GH_NT) void ghead_bwd_slices_kernel(GHeadArgs g) {
  extern __shared__ float4 gh_smem4[];
  float* const sm = reinterpret_cast<float*>(gh_smem4);
  const advmil_ghead_t& a = g.a;
  const int B = a.B, d0 = a.d0, d2 = a.d2;
  const bool two = a.d1 > 0;
  const int ds = two ? a.d1 : a.d2;
  const float* const Ws = two ? a.Wr : a.W0;
  float* const dWs = two ? a.dWr : a.dW0;
  float* const dbs = two ? a.dbr : a.db0;
  const float ps = two ? a.p1 : a.p2;
  const int tid = threadIdx.x, j = blockIdx.x, c0 = j * GH_CW;
  const int PX = d0 + 4, q0 = d0 >> 2, P2 = d2 + 4;
  float* const sX = sm;                       // [B][PX]
  float* const sW = sX + B * PX;              // [16][PX]   slice-layer weight rows c0 .. c0 + 15
  float* const sHs = sW + GH_CW * PX;         // [32][16]   saved slice-layer activations
  float* const sGs = sHs + 32 * GH_CW;        // [32][16]   gradient at the slice layer's pre-activation
  float* const sDz = sGs + 32 * GH_CW;        // [32]
  float* const sG2 = sDz + 32;                // [B][P2]    (two) gradient at the second layer's pre-activation
  float* const sW0 = sG2 + (two ? B * P2 : 0);   // [d2][16]  (two) W0[:, c0 .. c0 + 15]
  // ---- global reads
  for (int o = tid; o < B * q0; o += GH_NT) {
    const int b = o / q0, c = o - b * q0;
    *reinterpret_cast<float4*>(sX + b * PX + 4 * c) = *reinterpret_cast<const float4*>(a.x + (int64_t)b * a.ldx + 4 * c);
  }
  for (int o = tid; o < GH_CW * q0; o += GH_NT) {
    const int r = o / q0, c = o - r * q0;
    *reinterpret_cast<float4*>(sW + r * PX + 4 * c) = *reinterpret_cast<const float4*>(Ws + (int64_t)(c0 + r) * d0 + 4 * c);
  }
  for (int o = tid; o < B * 4; o += GH_NT) {
    const int b = o >> 2, q = o & 3;
    *reinterpret_cast<float4*>(sHs + b * GH_CW + 4 * q) = *reinterpret_cast<const float4*>(a.hs + (int64_t)b * ds + c0 + 4 * q);
  }
  if (two)
    for (int o = tid; o < d2 * 4; o += GH_NT) {
      const int n = o >> 2, q = o & 3;
      *reinterpret_cast<float4*>(sW0 + n * GH_CW + 4 * q) = *reinterpret_cast<const float4*>(a.W0 + (int64_t)n * ds + c0 + 4 * q);
    }
  if (tid < B) {
    const float p = a.pred[tid];
    sDz[tid] = a.dpred[tid] * (a.out_act ? p * (1.0f - p) : 1.0f);
  }
  LDS_BARRIER();
  const float inv2 = (a.seed && a.p2 > 0.f) ? hw_rcp(1.0f - a.p2) : 1.f;
  const float invs = (a.seed && ps > 0.f) ? hw_rcp(1.0f - ps) : 1.f;
  const int c = tid & 15, bq = tid >> 4;
  if (two) {
    // gradient at the second layer's pre-activation, all of it (every workgroup): g2[b][n] = dz[b] W1[n] [h2 > 0] / (1 - p2)
    for (int o0 = tid; o0 < B * d2; o0 += 4 * GH_NT) {
      float hv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) hv[u] = o0 + u * GH_NT < B * d2 ? a.h2[o0 + u * GH_NT] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = o0 + u * GH_NT;
        if (o < B * d2) {
          const int b = o / d2, n = o - b * d2;
          sG2[b * P2 + n] = hv[u] > 0.f ? sDz[b] * a.W1[n] * inv2 : 0.f;
        }
      }
    }
    LDS_BARRIER();
    for (int b = bq; b < B; b += 16) {
      float acc0 = 0.f, acc1 = 0.f;
      const float* gr = sG2 + b * P2;
      for (int n = 0; n < d2; n += 2) {
        acc0 += gr[n] * sW0[n * GH_CW + c];
        acc1 += gr[n + 1] * sW0[(n + 1) * GH_CW + c];
      }
      sGs[b * GH_CW + c] = sHs[b * GH_CW + c] > 0.f ? (acc0 + acc1) * invs : 0.f;
    }
  } else {
    for (int b = bq; b < B; b += 16) sGs[b * GH_CW + c] = sHs[b * GH_CW + c] > 0.f ? sDz[b] * a.W1[c0 + c] * invs : 0.f;
  }
  LDS_BARRIER();
  // ---- second layer's weight gradient, this slice's columns: dW0[n][c0 + c] += sum_b g2[b][n] hs[b][c]
  // (the destinations' old values are requested FIRST, all of them, and added at the end: read-add-write element by element is one
  // memory round trip per element -- the compiler cannot reorder the loads across the stores to the same array)
  if (two && a.dW0) {
    float old[16], acc[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int n = bq + 16 * it;
      old[it] = n < d2 ? a.dW0[(int64_t)n * ds + c0 + c] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int n = bq + 16 * it;
      float t = 0.f;
      if (n < d2)
        for (int b = 0; b < B; ++b) t += sG2[b * P2 + n] * sHs[b * GH_CW + c];
      acc[it] = t;
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int n = bq + 16 * it;
      if (n < d2) a.dW0[(int64_t)n * ds + c0 + c] = old[it] + acc[it];
    }
  }
  // ---- slice layer's weight gradient rows: dWs[c0 + c][k] += sum_b gs[b][c] X[b][k]; bias
  if (dWs) {
    for (int k4 = tid; k4 < q0; k4 += GH_NT) {
      float4 acc[GH_CW], old[GH_CW];
#pragma unroll
      for (int cc = 0; cc < GH_CW; ++cc) {
        acc[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
        old[cc] = *reinterpret_cast<const float4*>(dWs + (int64_t)(c0 + cc) * d0 + 4 * k4);
      }
      for (int b = 0; b < B; ++b) {
        const float4 x = *reinterpret_cast<const float4*>(sX + b * PX + 4 * k4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 gq = *reinterpret_cast<const float4*>(sGs + b * GH_CW + 4 * q);
          const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float4& t = acc[4 * q + u];
            t.x += gv[u] * x.x; t.y += gv[u] * x.y; t.z += gv[u] * x.z; t.w += gv[u] * x.w;
          }
        }
      }
#pragma unroll
      for (int cc = 0; cc < GH_CW; ++cc) {
        float4 o = old[cc];
        o.x += acc[cc].x; o.y += acc[cc].y; o.z += acc[cc].z; o.w += acc[cc].w;
        *reinterpret_cast<float4*>(dWs + (int64_t)(c0 + cc) * d0 + 4 * k4) = o;
      }
    }
  }
  if (dbs && tid < GH_CW) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += sGs[b * GH_CW + tid];
    dbs[c0 + tid] += acc;
  }
  // ---- this slice's share of dX: Q_j[b][k] = sum_c gs[b][c] Ws[c0 + c][k]
  if (a.dx) {
    for (int k4 = tid; k4 < q0; k4 += GH_NT) {
      float4 w[GH_CW];
#pragma unroll
      for (int cc = 0; cc < GH_CW; ++cc) w[cc] = *reinterpret_cast<const float4*>(sW + cc * PX + 4 * k4);
      for (int b = 0; b < B; ++b) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 gq = *reinterpret_cast<const float4*>(sGs + b * GH_CW + 4 * q);
          const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 ww = w[4 * q + u];
            acc.x += gv[u] * ww.x; acc.y += gv[u] * ww.y; acc.z += gv[u] * ww.z; acc.w += gv[u] * ww.w;
          }
        }
        *reinterpret_cast<float4*>(a.ws + ((int64_t)j * B + b) * d0 + 4 * k4) = acc;
      }
    }
  }
  // ---- workgroup 0: the output layer's gradients and the second layer's bias gradient
  if (j == 0) {
    const uint64_t keyn = a.noise_mode == 3 ? rng_key(*a.seed, a.sid_noise) : 0;
    for (int n = tid; n < d2; n += GH_NT) {
      const float* hl = two ? a.h2 : a.hs;       // the layer the output layer reads: [B, d2]
      float gw = 0.f, gn = 0.f, gb = 0.f;
      for (int b0 = 0; b0 < B; b0 += 8) {
        float hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hv[u] = b0 + u < B ? hl[(int64_t)(b0 + u) * d2 + n] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + u;
          if (b < B) {
            gw += sDz[b] * hv[u];
            if (a.noise_mode >= 2) gn += sDz[b] * gh_noise(a, keyn, b, n);
            if (two) gb += sG2[b * P2 + n];
          }
        }
      }
      if (a.dW1) {
        a.dW1[n] += gw;
        if (a.noise_mode >= 2) a.dW1[d2 + n] += gn;
      }
      if (two && a.db0) a.db0[n] += gb;
    }
    if (tid == 0 && a.db1) {
      float acc = 0.f;
      for (int b = 0; b < B; ++b) acc += sDz[b];
      a.db1[0] += acc;
    }
  }
}

// backward D: one workgroup per bag sums the dX shares
__global__ __launch_bounds__(128) void ghead_bwd_finish_kernel(GHeadArgs g) {
  const advmil_ghead_t& a = g.a;
  const int B = a.B, d0 = a.d0, b = blockIdx.x, nw = g.nw;
  for (int k4 = threadIdx.x; k4 < (d0 >> 2); k4 += 128) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < nw; j += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = j + u < nw ? *reinterpret_cast<const float4*>(a.ws + ((int64_t)(j + u) * B + b) * d0 + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(a.dx + (int64_t)b * a.lddx + 4 * k4) = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
static int gh_ds(const advmil_ghead_t* a) { return a->d1 > 0 ? a->d1 : a->d2; }

static int gh_check(const advmil_ghead_t* a) {
  if (!a || a->B < 1 || a->B > 32 || a->d0 < 4 || a->d0 > 512 || (a->d0 & 3) || a->d1 < 0 || a->d2 < 4 || a->d2 > 256 || (a->d2 & 3)) return ADVMIL_EINVAL;
  const int ds = gh_ds(a);
  if ((ds % GH_CW) || ds > 1024 || ds / GH_CW > GH_NT) return ADVMIL_EINVAL;
  if (!a->x || a->ldx < a->d0 || (a->ldx & 3) || ((uintptr_t)a->x & 15) || !a->W0 || !a->W1 || !a->hs || !a->pred || !a->ws) return ADVMIL_EINVAL;
  if (a->d1 > 0 && (!a->Wr || !a->h2 || ((uintptr_t)a->Wr & 15))) return ADVMIL_EINVAL;
  if (((uintptr_t)a->W0 & 15) || ((uintptr_t)a->hs & 15) || ((uintptr_t)a->ws & 15)) return ADVMIL_EINVAL;
  if (a->noise_mode < 0 || a->noise_mode > 3 || (a->noise_mode == 2 && !a->noise) || (a->noise_mode == 3 && !a->seed)) return ADVMIL_EINVAL;
  if (!(a->p1 >= 0.f && a->p1 < 1.f) || !(a->p2 >= 0.f && a->p2 < 1.f)) return ADVMIL_EINVAL;
  if ((a->p1 > 0.f || a->p2 > 0.f) && !a->seed) return ADVMIL_EINVAL;
  return ADVMIL_OK;
}

extern "C" size_t advmil_ghead_workspace_bytes(int B, int d0, int d1, int d2) {
  if (B < 1 || d0 < 1 || d2 < 1) return 0;
  const int ds = d1 > 0 ? d1 : d2;
  const size_t nw = (size_t)(ds + GH_CW - 1) / GH_CW;
  const size_t wide = (size_t)(d0 > d2 ? d0 : d2);
  return nw * (size_t)B * wide * sizeof(float);
}

static size_t gh_lds_fwd(const advmil_ghead_t* a) { return ((size_t)(a->B + GH_CW) * (a->d0 + 4) + 32 * GH_CW) * sizeof(float); }
static size_t gh_lds_bwd(const advmil_ghead_t* a) {
  size_t f = (size_t)(a->B + GH_CW) * (a->d0 + 4) + 2 * 32 * GH_CW + 32;
  if (a->d1 > 0) f += (size_t)a->B * (a->d2 + 4) + (size_t)a->d2 * GH_CW;
  return f * sizeof(float);
}

static int gh_lds_attr() {      // once: both slice kernels may ask for more than the 64 KB default of dynamic LDS
  static const int rc = []() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ghead_fwd_slices_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(ghead_bwd_slices_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return (int)e;
  }();
  return rc;
}

extern "C" int advmil_ghead_fwd(const advmil_ghead_t* a, advmil_stream_t stream_) {
  const int rc = gh_check(a);
  if (rc) return rc;
  if (a->ws_bytes < advmil_ghead_workspace_bytes(a->B, a->d0, a->d1, a->d2)) return ADVMIL_EWORKSPACE;
  const size_t lds = gh_lds_fwd(a);
  if (lds > 160 * 1024) return ADVMIL_EINVAL;
  if (gh_lds_attr()) return ADVMIL_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GHeadArgs g;
  g.a = *a;
  g.nw = gh_ds(a) / GH_CW;
  hipLaunchKernelGGL(ghead_fwd_slices_kernel, dim3(g.nw), dim3(GH_NT), lds, stream, g);
  ADVMIL_LAUNCH_CHECK();
  hipLaunchKernelGGL(ghead_fwd_finish_kernel, dim3(a->B), dim3(GH_NT), 0, stream, g);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_ghead_bwd(const advmil_ghead_t* a, advmil_stream_t stream_) {
  const int rc = gh_check(a);
  if (rc) return rc;
  if (!a->dpred) return ADVMIL_EINVAL;
  if (a->dx && (a->lddx < a->d0 || (a->lddx & 3) || ((uintptr_t)a->dx & 15))) return ADVMIL_EINVAL;
  if (a->ws_bytes < advmil_ghead_workspace_bytes(a->B, a->d0, a->d1, a->d2)) return ADVMIL_EWORKSPACE;
  const float* grads[] = {a->dWr, a->dW0};
  for (const float* p : grads)
    if ((uintptr_t)p & 15) return ADVMIL_EINVAL;
  const size_t lds = gh_lds_bwd(a);
  if (lds > 160 * 1024) return ADVMIL_EINVAL;
  if (gh_lds_attr()) return ADVMIL_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  GHeadArgs g;
  g.a = *a;
  g.nw = gh_ds(a) / GH_CW;
  hipLaunchKernelGGL(ghead_bwd_slices_kernel, dim3(g.nw), dim3(GH_NT), lds, stream, g);
  ADVMIL_LAUNCH_CHECK();
  if (a->dx) {
    hipLaunchKernelGGL(ghead_bwd_finish_kernel, dim3(a->B), dim3(128), 0, stream, g);
    ADVMIL_LAUNCH_CHECK();
  }
  return ADVMIL_OK;
}
