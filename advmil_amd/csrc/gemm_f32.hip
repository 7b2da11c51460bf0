// fp32 dense contraction engine for gfx950: v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak).
//
// Block tile 128x128x32, 256 threads = 4 waves in a 2x2 grid, each wave a 64x64 sub-tile held
// as 2x2 MFMA 32x32 accumulators (64 VGPRs). Operands are staged global -> registers -> LDS
// (register prefetch of the next k-chunk overlaps the 64 MFMAs of the current one).
//
// Two LDS images, chosen per operand by how the SOURCE is laid out:
//   k-contiguous source ([rows,K] row-major, e.g. X or a Linear weight): LDS [row][36] (32 k + 4 pad),
//     one ds_read_b128 per 32-row fragment per 8 k; pitch 36 makes every 16-lane b128 group hit 64
//     distinct banks.
//   m-contiguous source ([K,rows], e.g. dY in dW = dY^T X): LDS [k][128], ds_read_b32 (lanes read
//     32 consecutive floats -> conflict free).
// The MFMA k-slot of lane-half `hi` at step (t4,u) is k = 8*t4 + 4*hi + u for BOTH operands; the
// contraction order is a permutation of 0..31, which fp32 accumulation does not care about.
#include "common.h"
#include "../../include/advmil_hip.h"

#define BM 128
#define BN 128
#define BK 32
#define PITCH_KC 36
#define PITCH_MC 128

struct GemmArgs {
  int64_t M, N, K;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int64_t k_chunk;   // K range per split (multiple of BK)
  float* ws;         // [splits][M][N] partials when splits > 1
  int splits;
  int mtiles;
  advmil_epilogue_t epi;
};

__device__ __forceinline__ float epilogue_elem(const advmil_epilogue_t& e, float acc, int64_t m, int64_t n, int64_t N,
                                               uint64_t key, float inv_keep) {
  float v = acc * e.alpha;
  if (e.bias) v += e.bias[n];
  if (e.rowv) v += e.rowv[m] * e.colv[n];
  v = act_apply(n < e.act_split ? e.act0 : e.act1, v);
  if (e.seed && e.drop_p > 0.0f) v *= rng_keep(key, (uint64_t)(m * N + n), e.drop_p, inv_keep);
  if (e.maskref) v *= (e.maskref[m * (int64_t)e.ldmask + n] > 0.0f ? e.mask_scale : 0.0f);
  return v;
}

template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t rows,
                                          int64_t k0, int64_t kend, int tid, float4 (&r)[4]) {
  if (KC) {
    // 128 rows x 32 k; thread -> (row = p*32 + tid/8, k = (tid%8)*4)
    const int kq = (tid & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t row = row0 + p * 32 + (tid >> 3);
      const int64_t k = k0 + kq;
      if (row < rows && k < kend)
        r[p] = *reinterpret_cast<const float4*>(src + row * ld + k);
      else
        r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    // 32 k x 128 rows(m); thread -> (k = p*8 + tid/32, m = (tid%32)*4)
    const int mq = (tid & 31) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t k = k0 + p * 8 + (tid >> 5);
      const int64_t m = row0 + mq;
      if (k < kend && m < rows)
        r[p] = *reinterpret_cast<const float4*>(src + k * ld + m);
      else
        r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ s, int tid, const float4 (&r)[4]) {
  if (KC) {
    const int kq = (tid & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(s + (p * 32 + (tid >> 3)) * PITCH_KC + kq) = r[p];
  } else {
    const int mq = (tid & 31) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(s + (p * 8 + (tid >> 5)) * PITCH_MC + mq) = r[p];
  }
}

// fragment for the 32-row block starting at `rbase` (0..127), k-group t4: f[u], u = 0..3
template <bool KC>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int rbase, int t4, int i, int hi, float (&f)[4]) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4*>(s + (rbase + i) * PITCH_KC + t4 * 8 + hi * 4);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = s[(t4 * 8 + hi * 4 + u) * PITCH_MC + rbase + i];
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[BM * PITCH_KC];
  __shared__ __attribute__((aligned(16))) float sB[BN * PITCH_KC];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;

  const int bid = blockIdx.x;
  const int mt_i = bid % g.mtiles, nt_i = bid / g.mtiles;
  const int64_t m0 = (int64_t)mt_i * BM, n0 = (int64_t)nt_i * BN;
  const int z = blockIdx.y;
  const int64_t kbeg = (int64_t)z * g.k_chunk;
  const int64_t kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float4 ra[4], rb[4];
  if (kbeg < kend) {
    load_tile<A_KC>(g.A, g.lda, m0, g.M, kbeg, kend, tid, ra);
    load_tile<B_KC>(g.B, g.ldb, n0, g.N, kbeg, kend, tid, rb);
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // all waves finished reading the previous chunk
    store_tile<A_KC>(sA, tid, ra);
    store_tile<B_KC>(sB, tid, rb);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch next chunk; lands while the MFMAs below run
      load_tile<A_KC>(g.A, g.lda, m0, g.M, k0 + BK, kend, tid, ra);
      load_tile<B_KC>(g.B, g.ldb, n0, g.N, k0 + BK, kend, tid, rb);
    }
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      float fa[2][4], fb[2][4];
      read_frag<A_KC>(sA, wr * 64, t4, i, hi, fa[0]);
      read_frag<A_KC>(sA, wr * 64 + 32, t4, i, hi, fa[1]);
      read_frag<B_KC>(sB, wc * 64, t4, i, hi, fb[0]);
      read_frag<B_KC>(sB, wc * 64 + 32, t4, i, hi, fb[1]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][u], fb[0][u], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][u], fb[1][u], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][u], fb[0][u], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][u], fb[1][u], acc[1][1], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const advmil_epilogue_t& e = g.epi;
  const bool direct = (g.splits == 1);
  uint64_t key = 0;
  float inv_keep = 1.0f;
  if (direct && e.seed && e.drop_p > 0.0f) {
    key = rng_key(*e.seed, e.stream_id);
    inv_keep = 1.0f / (1.0f - e.drop_p);
  }
  float* ws = direct ? nullptr : g.ws + (int64_t)z * g.M * g.N;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t col = n0 + wc * 64 + b * 32 + i;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row >= g.M) continue;
        if (direct) {
          float v = epilogue_elem(e, acc[a][b][r], row, col, g.N, key, inv_keep);
          float* c = g.C + row * g.ldc + col;
          if (e.accumulate) v += *c;
          *c = v;
        } else {
          ws[row * g.N + col] = acc[a][b][r];
        }
      }
    }
  }
}

// split-K reduction + epilogue; one thread per 4 consecutive columns
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g) {
  const int64_t n4 = g.N / 4;
  const int64_t total = g.M * n4;
  const advmil_epilogue_t& e = g.epi;
  uint64_t key = 0;
  float inv_keep = 1.0f;
  if (e.seed && e.drop_p > 0.0f) {
    key = rng_key(*e.seed, e.stream_id);
    inv_keep = 1.0f / (1.0f - e.drop_p);
  }
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
      float o = epilogue_elem(e, v[q], m, n + q, g.N, key, inv_keep);
      if (e.accumulate) o += c[q];
      c[q] = o;
    }
  }
}

extern "C" size_t advmil_gemm_f32_workspace_bytes(int64_t M, int64_t N, int splits) {
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int advmil_gemm_f32(int a_kc, int b_kc, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                               const float* B, int64_t ldb, float* C, int64_t ldc, const advmil_epilogue_t* epi,
                               int splits, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !B || !C || !epi || M <= 0 || N <= 0 || K <= 0) return ADVMIL_EINVAL;
  if ((lda & 3) || (ldb & 3)) return ADVMIL_EINVAL;
  if (a_kc ? (K & 3) : (M & 3)) return ADVMIL_EINVAL;
  if (b_kc ? (K & 3) : (N & 3)) return ADVMIL_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return ADVMIL_EINVAL;
  if (splits < 1) splits = 1;
  const int64_t kchunks = (K + BK - 1) / BK;
  if (splits > kchunks) splits = (int)kchunks;
  if (splits > 1 && (N & 3)) return ADVMIL_EINVAL;
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.k_chunk = ((kchunks + splits - 1) / splits) * BK;
  splits = (int)((K + g.k_chunk - 1) / g.k_chunk);
  g.splits = splits;
  g.ws = (float*)ws;
  g.epi = *epi;
  if (splits > 1) {
    if (!ws || ws_bytes < advmil_gemm_f32_workspace_bytes(M, N, splits)) return ADVMIL_EWORKSPACE;
    if ((uintptr_t)ws & 15) return ADVMIL_EINVAL;
  }
  g.mtiles = (int)((M + BM - 1) / BM);
  const int ntiles = (int)((N + BN - 1) / BN);
  dim3 grid(g.mtiles * ntiles, splits), block(256);
  if (a_kc && b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, stream, g);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, stream, g);
  else if (!a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, stream, g);
  ADVMIL_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t total = M * (N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), block, 0, stream, g);
    ADVMIL_LAUNCH_CHECK();
  }
  return ADVMIL_OK;
}
