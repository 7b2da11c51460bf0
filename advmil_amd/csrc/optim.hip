// Flat-arena optimizer and utility kernels (HBM-bound, float4 streams).
#include "common.h"
#include "../../include/advmil_hip.h"

extern "C" int advmil_version(void) { return 100; }

// torch.optim.Adam (L2-in-grad) with the L1 sub-gradient of loss_reg_l1 folded in.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ wd, int64_t n, float lr,
                                                   float b1, float b2, float eps, float gscale, float l1,
                                                   const int32_t* __restrict__ step) {
  const int t = *step + 1;   // the launcher bumps *step after this kernel
  const float bc1 = 1.f - powf(b1, (float)t);
  const float bc2 = 1.f - powf(b2, (float)t);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float w = p[i];
    float g = grad[i] * gscale;
    if (l1 != 0.f) g += l1 * (w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f));
    if (wd) g += wd[i] * w;
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    p[i] = w - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}
__global__ void step_inc_kernel(int32_t* step) { *step += 1; }

extern "C" int advmil_adam_step(float* p, const float* grad, float* m, float* v, const float* wd, int64_t n, float lr,
                                float beta1, float beta2, float eps, float grad_scale, float l1_coef, int32_t* step,
                                advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!p || !grad || !m || !v || !step || n <= 0) return ADVMIL_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, grad, m, v, wd, n, lr, beta1, beta2, eps, grad_scale,
                     l1_coef, step);
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, stream, step);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

#define ABS_BLOCKS 256
__global__ __launch_bounds__(256) void abs_sum_partial_kernel(const float* __restrict__ p, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += fabsf(p[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}
extern "C" size_t advmil_abs_sum_workspace_bytes(int64_t n) { (void)n; return ABS_BLOCKS * sizeof(float); }
extern "C" int advmil_abs_sum(const float* p, int64_t n, float* out, void* ws, size_t ws_bytes, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!p || !out || !ws || n <= 0) return ADVMIL_EINVAL;
  if (ws_bytes < advmil_abs_sum_workspace_bytes(n)) return ADVMIL_EWORKSPACE;
  hipLaunchKernelGGL(abs_sum_partial_kernel, dim3(ABS_BLOCKS), dim3(256), 0, stream, p, n, (float*)ws);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, ABS_BLOCKS, out);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

__global__ __launch_bounds__(256) void uniform_fill_kernel(float* __restrict__ out, int64_t n, const uint64_t* seed, uint64_t stream_id) {
  const uint64_t key = rng_key(*seed, stream_id);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = rng_uniform(key, (uint64_t)i);
}
extern "C" int advmil_uniform_fill(float* out, int64_t n, const uint64_t* seed, uint64_t stream_id, advmil_stream_t stream_) {
  if (!out || !seed || n <= 0) return ADVMIL_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(uniform_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, out, n, seed, stream_id);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// y[i] = x[i] * (u(seed, stream, i) >= p ? 1/(1-p) : 0): the dropout of the [B, d]-sized head tensors (and, applied to dy, its
// backward) as ONE launch; same draw as the GEMM epilogue's dropout and synth.dropout_keep.
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p,
                                                            const uint64_t* seed, uint64_t stream_id) {
  const uint64_t key = rng_key(*seed, stream_id);
  const float inv = 1.0f / (1.0f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * rng_keep(key, (uint64_t)i, p, inv);
}
extern "C" int advmil_dropout_apply(const float* x, float* y, int64_t n, float p, const uint64_t* seed, uint64_t stream_id,
                                    advmil_stream_t stream_) {
  if (!x || !y || !seed || n <= 0 || !(p >= 0.0f && p < 1.0f)) return ADVMIL_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, y, n, p, seed, stream_id);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

__global__ void seed_advance_kernel(uint64_t* seed, uint64_t inc) { *seed += inc; }
extern "C" int advmil_seed_advance(uint64_t* seed, uint64_t inc, advmil_stream_t stream_) {
  if (!seed) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, seed, inc);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
