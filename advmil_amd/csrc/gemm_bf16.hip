// bf16-operand / fp32-accumulate contraction engine for gfx950: v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense peak).
//
// Only the "NT" form exists: C[M,N] = epi(alpha * A[M,K] . B[N,K]^T) with BOTH operands k-contiguous bf16. The k-strided
// contractions of the backward pass (dW = dY^T X, dX = dY W) are brought to this form by having the producing kernels emit a
// TRANSPOSED bf16 copy of their output next to the fp32 one (this kernel's `Ct`, advmil_cast_bf16's `dstT`), which costs
// 2 bytes per element of extra HBM writes and avoids 2-byte scattered LDS traffic that would starve a 16x-faster MFMA.
//
// Block tile (64*TM) x (64*TN) x 64, 4 waves (2x2), each wave TM x TN accumulators of 32x32 (fp32, C/D layout identical to the
// fp32 MFMA). LDS image [row][72] bf16 (64 k + 8 pad = 144 B pitch = 36 dwords: every 16-lane ds_read_b128 group covers 64
// distinct banks). Lane (i = l&31, hi = l>>5) feeds the MFMA 8 consecutive k at k0 + 8*hi for row i -- the same slot map for
// A and B, so the contraction is exact whatever order the hardware walks the slots in.
#include "common.h"
#include "../../include/advmil_hip.h"

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag16 { uint4 u; bf16x8 v; };

#define BKH 64
#define PITCH_H 72   // bf16 elements per LDS row

__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even, NaN-safe enough for finite data
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

struct GemmHArgs {
  int64_t M, N, K;
  const bf16_t* A; int64_t lda;
  const bf16_t* B; int64_t ldb;
  float* C; int64_t ldc;          // fp32 result (may be NULL)
  bf16_t* Cb; int64_t ldcb;       // bf16 copy, row-major (may be NULL)
  bf16_t* Ct; int64_t ldct;       // bf16 copy, TRANSPOSED [N][ldct] (may be NULL)
  int64_t k_chunk; float* ws; int splits; int mtiles, ntiles;
  advmil_epilogue_t epi;
};

__device__ __forceinline__ float epilogue_elem_h(const advmil_epilogue_t& e, float acc, int64_t m, int64_t n, int64_t N,
                                                 uint64_t key, float inv_keep) {
  float v = acc * e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.rowv) v += e.rowv[m] * e.colv[(e.rowseg ? (int64_t)e.rowseg[m] * N : 0) + n];
  v = act_apply(n < e.act_split ? e.act0 : e.act1, v);
  if (e.seed && e.drop_p > 0.0f) v *= rng_keep(key, (uint64_t)(m * N + n), e.drop_p, inv_keep);
  if (e.maskref) v *= (e.maskref[m * (int64_t)e.ldmask + n] > 0.0f ? e.mask_scale : 0.0f);
  return v;
}

template <int ROWS>
__device__ __forceinline__ void load_tile_h(const bf16_t* __restrict__ src, int64_t ld, int64_t row0, int64_t rows, int64_t k0,
                                            int64_t kend, int tid, uint4 (&r)[ROWS / 32]) {
#pragma unroll
  for (int p = 0; p < ROWS / 32; ++p) {
    const int e = p * 256 + tid;
    const int64_t row = row0 + (e >> 3);
    const int64_t k = k0 + (e & 7) * 8;
    r[p] = (row < rows && k < kend) ? *reinterpret_cast<const uint4*>(src + row * ld + k) : make_uint4(0, 0, 0, 0);
  }
}

template <int ROWS>
__device__ __forceinline__ void store_tile_h(bf16_t* __restrict__ s, int tid, const uint4 (&r)[ROWS / 32]) {
#pragma unroll
  for (int p = 0; p < ROWS / 32; ++p) {
    const int e = p * 256 + tid;
    *reinterpret_cast<uint4*>(s + (e >> 3) * PITCH_H + (e & 7) * 8) = r[p];
  }
}

template <int TM, int TN>
__global__ __launch_bounds__(256, (TM * TN > 4 ? 1 : 2)) void gemm_bf16_nt_kernel(GemmHArgs g) {
  constexpr int BM_ = 64 * TM, BN_ = 64 * TN;
  __shared__ __attribute__((aligned(16))) bf16_t smem[(BM_ + BN_) * PITCH_H];
  bf16_t* const sA = smem;
  bf16_t* const sB = smem + BM_ * PITCH_H;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware tile order (see gemm_f32.hip)
  const int bid = blockIdx.x;
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
  const int z = blockIdx.y;
  const int64_t kbeg = (int64_t)z * g.k_chunk;
  const int64_t kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  uint4 ra[BM_ / 32], rb[BN_ / 32];
  if (kbeg < kend) {
    load_tile_h<BM_>(g.A, g.lda, m0, g.M, kbeg, kend, tid, ra);
    load_tile_h<BN_>(g.B, g.ldb, n0, g.N, kbeg, kend, tid, rb);
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BKH) {
    __syncthreads();
    store_tile_h<BM_>(sA, tid, ra);
    store_tile_h<BN_>(sB, tid, rb);
    __syncthreads();
    if (k0 + BKH < kend) {
      load_tile_h<BM_>(g.A, g.lda, m0, g.M, k0 + BKH, kend, tid, ra);
      load_tile_h<BN_>(g.B, g.ldb, n0, g.N, k0 + BKH, kend, tid, rb);
    }
#pragma unroll
    for (int ks = 0; ks < BKH / 16; ++ks) {
      Frag16 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[a].u = *reinterpret_cast<const uint4*>(sA + (wr * 32 * TM + a * 32 + i) * PITCH_H + ks * 16 + hi * 8);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        fb[b].u = *reinterpret_cast<const uint4*>(sB + (wc * 32 * TN + b * 32 + i) * PITCH_H + ks * 16 + hi * 8);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0);
    }
  }

  // ---- epilogue through a per-wave fp32 LDS patch [32][36]
  const advmil_epilogue_t& e = g.epi;
  const bool direct = (g.splits == 1);
  uint64_t key = 0;
  float inv_keep = 1.0f;
  if (direct && e.seed && e.drop_p > 0.0f) {
    key = rng_key(*e.seed, e.stream_id);
    inv_keep = 1.0f / (1.0f - e.drop_p);
  }
  float* const outf = direct ? g.C : g.ws + (int64_t)z * g.M * g.N;
  const int64_t ldo = direct ? g.ldc : g.N;
  const bool vec_ok = outf && ((ldo & 3) == 0) && (((uintptr_t)outf & 15) == 0);
  float* const patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
  __syncthreads();
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + i] = acc[a][b][r];
      __syncthreads();
      const int64_t rbase = m0 + wr * 32 * TM + a * 32;
      const int64_t cbase = n0 + wc * 32 * TN + b * 32;
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const int idx = q * 64 + lane;
        const int pr = idx >> 3, pc = (idx & 7) * 4;
        const int64_t row = rbase + pr, col = cbase + pc;
        const float4 v4 = *reinterpret_cast<const float4*>(patch + pr * 36 + pc);
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        if (row < g.M && col < g.N) {
          const int nvalid = (g.N - col >= 4) ? 4 : (int)(g.N - col);
          if (direct) {
            float* c = g.C ? g.C + row * g.ldc + col : nullptr;
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (t < nvalid) {
                v[t] = epilogue_elem_h(e, v[t], row, col + t, g.N, key, inv_keep);
                if (e.accumulate && c) v[t] += c[t];
              }
            if (g.Ct) *reinterpret_cast<float4*>(patch + pr * 36 + pc) = make_float4(v[0], v[1], v[2], v[3]);
            if (g.Cb) {
              bf16_t* cb = g.Cb + row * g.ldcb + col;
              if (nvalid == 4 && (g.ldcb & 3) == 0) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(cb) = pk;
              } else {
                for (int t = 0; t < nvalid; ++t) cb[t] = f2bf(v[t]);
              }
            }
          }
          if (outf) {
            float* c = outf + row * ldo + col;
            if (nvalid == 4 && vec_ok) *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int t = 0; t < nvalid; ++t) c[t] = v[t];
          }
        }
      }
      if (direct && g.Ct) {
        // transposed bf16 copy: lane owns output row (= source column) c = lane&31 and 16 consecutive source rows
        __syncthreads();
        const int c = lane & 31, r0 = (lane >> 5) * 16;
        const int64_t col = cbase + c;
        if (col < g.N) {
          bf16_t* dst = g.Ct + col * g.ldct + rbase + r0;
          if (rbase + r0 + 16 <= g.M && (g.ldct & 7) == 0) {
            uint32_t w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
              w[t] = (uint32_t)f2bf(patch[(r0 + 2 * t) * 36 + c]) | ((uint32_t)f2bf(patch[(r0 + 2 * t + 1) * 36 + c]) << 16);
            *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(dst + 8) = make_uint4(w[4], w[5], w[6], w[7]);
          } else {
            for (int t = 0; t < 16; ++t)
              if (rbase + r0 + t < g.M) dst[t] = f2bf(patch[(r0 + t) * 36 + c]);
          }
        }
      }
      __syncthreads();
    }
  }
}

// split-K reduction for the bf16 engine: fp32 partials -> fp32 C with the epilogue (no bf16 copies on this path)
__global__ __launch_bounds__(256) void gemm_bf16_splitk_reduce_kernel(GemmHArgs g) {
  const int64_t n4 = g.N / 4;
  const int64_t total = g.M * n4;
  const advmil_epilogue_t& e = g.epi;
  uint64_t key = 0;
  float inv_keep = 1.0f;
  if (e.seed && e.drop_p > 0.0f) { key = rng_key(*e.seed, e.stream_id); inv_keep = 1.0f / (1.0f - e.drop_p); }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / n4, n = (idx % n4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < g.splits; ++z) {
      const float4 p = *reinterpret_cast<const float4*>(g.ws + ((int64_t)z * g.M + m) * g.N + n);
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    float v[4] = {s.x, s.y, s.z, s.w};
    float* c = g.C + m * g.ldc + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o = epilogue_elem_h(e, v[q], m, n + q, g.N, key, inv_keep);
      if (e.accumulate) o += c[q];
      c[q] = o;
    }
  }
}

static int64_t n_tiles_h(int tile, int64_t M, int64_t N) {
  const int tm = tile / 10, tn = tile % 10;
  return ((M + 64 * tm - 1) / (64 * tm)) * ((N + 64 * tn - 1) / (64 * tn));
}

extern "C" int advmil_gemm_bf16_plan(int64_t M, int64_t N, int64_t K, int* tile, int* splits) {
  static const int order[4] = {22, 12, 11, 11};
  for (int c = 0; c < 3; ++c)
    if (n_tiles_h(order[c], M, N) >= 512) { *tile = order[c]; *splits = 1; return ADVMIL_OK; }
  const int64_t w = n_tiles_h(22, M, N);
  if (K >= 8192 && w < 256 && (N & 3) == 0) {          // deep-K weight-gradient contractions
    int64_t sp = (512 + w - 1) / w;
    const int64_t cap = K / 2048 > 0 ? K / 2048 : 1;
    if (sp > cap) sp = cap;
    *tile = 22; *splits = (int)sp;
    return ADVMIL_OK;
  }
  *tile = n_tiles_h(22, M, N) >= 128 ? 22 : 11; *splits = 1;
  return ADVMIL_OK;
}

extern "C" size_t advmil_gemm_bf16_workspace_bytes(int64_t M, int64_t N, int splits) {
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

template <int TM, int TN>
static void launch_h(dim3 grid, hipStream_t stream, const GemmHArgs& g) {
  hipLaunchKernelGGL((gemm_bf16_nt_kernel<TM, TN>), grid, dim3(256), 0, stream, g);
}

extern "C" int advmil_gemm_bf16_nt(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                                   float* C, int64_t ldc, void* Cb, int64_t ldcb, void* Ct, int64_t ldct,
                                   const advmil_epilogue_t* epi, int splits, int tile, void* ws, size_t ws_bytes,
                                   advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !B || !epi || (!C && !Cb && !Ct) || M <= 0 || N <= 0 || K <= 0) return ADVMIL_EINVAL;
  if ((lda & 7) || (ldb & 7) || (K & 7)) return ADVMIL_EINVAL;              // 16 B = 8 bf16 operand loads
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ADVMIL_EINVAL;
  if (Ct && (((uintptr_t)Ct & 15))) return ADVMIL_EINVAL;
  if (splits < 1) splits = 1;
  const int64_t kchunks = (K + BKH - 1) / BKH;
  if (splits > kchunks) splits = (int)kchunks;
  if (splits > 1 && (!C || Cb || Ct || (N & 3))) return ADVMIL_EINVAL;
  GemmHArgs g;
  g.M = M; g.N = N; g.K = K; g.A = (const bf16_t*)A; g.lda = lda; g.B = (const bf16_t*)B; g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.Cb = (bf16_t*)Cb; g.ldcb = ldcb; g.Ct = (bf16_t*)Ct; g.ldct = ldct;
  g.k_chunk = ((kchunks + splits - 1) / splits) * BKH;
  splits = (int)((K + g.k_chunk - 1) / g.k_chunk);
  g.splits = splits; g.ws = (float*)ws; g.epi = *epi;
  if (splits > 1) {
    if (!ws || ws_bytes < advmil_gemm_bf16_workspace_bytes(M, N, splits)) return ADVMIL_EWORKSPACE;
    if ((uintptr_t)ws & 15) return ADVMIL_EINVAL;
  }
  if (tile == 0) { int t = 0, sp = 0; advmil_gemm_bf16_plan(M, N, K, &t, &sp); tile = t; }
  const int tm = tile / 10, tn = tile % 10;
  g.mtiles = (int)((M + 64 * tm - 1) / (64 * tm));
  g.ntiles = (int)((N + 64 * tn - 1) / (64 * tn));
  dim3 grid(g.mtiles * g.ntiles, splits);
  switch (tile) {
    case 22: launch_h<2, 2>(grid, stream, g); break;
    case 12: launch_h<1, 2>(grid, stream, g); break;
    case 11: launch_h<1, 1>(grid, stream, g); break;
    default: return ADVMIL_EINVAL;
  }
  ADVMIL_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t total = M * (N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_bf16_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g);
    ADVMIL_LAUNCH_CHECK();
  }
  return ADVMIL_OK;
}

// fp32 [R, C] -> bf16 [R, C] (dst, may be NULL) and/or bf16 TRANSPOSED [C, R] (dstT, may be NULL); 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int64_t lds_, int64_t R, int64_t C,
                                                        bf16_t* __restrict__ dst, int64_t ldd, bf16_t* __restrict__ dstT,
                                                        int64_t ldt) {
  __shared__ float tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int rr = ty; rr < 64; rr += 4) {
    const int64_t r = r0 + rr, c = c0 + tx;
    const float v = (r < R && c < C) ? src[r * lds_ + c] : 0.f;
    tile[rr][tx] = v;
    if (dst && r < R && c < C) dst[r * ldd + c] = f2bf(v);
  }
  if (!dstT) return;
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int64_t c = c0 + cc, r = r0 + tx;
    if (c < C && r < R) dstT[c * ldt + r] = f2bf(tile[tx][cc]);
  }
}

extern "C" int advmil_cast_bf16(const float* src, int64_t ld_src, int64_t R, int64_t C, void* dst, int64_t ld_dst, void* dstT,
                                int64_t ld_dstT, advmil_stream_t stream) {
  if (!src || (!dst && !dstT) || R <= 0 || C <= 0) return ADVMIL_EINVAL;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  hipLaunchKernelGGL(cast_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, R, C, (bf16_t*)dst, ld_dst,
                     (bf16_t*)dstT, ld_dstT);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
