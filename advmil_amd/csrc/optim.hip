// Flat-arena optimizer and utility kernels (HBM-bound, float4 streams).
#include "common.h"
#include "../../include/advmil_hip.h"

extern "C" int advmil_version(void) { return 100; }

// torch.optim.Adam (L2-in-grad) with the L1 sub-gradient of loss_reg_l1 folded in.
// abs_partial (optional): abs_partial[block] = sum |w| over the block's elements BEFORE the update -- the value of the L1 term the step
// logs (loss/utils.py:6-14) without a pass of its own over the arena. clear != 0: the gradient is zeroed behind its last read, so the next
// step needs no fill launch (graph replay only: p.grad reads zero afterwards).
__device__ __forceinline__ float adam_elem(float w, float g0, float& mi, float& vi, float wdv, float lr_step, float b1, float b2, float eps,
                                           float gscale, float l1, float inv_sqrt_bc2) {
  float g = g0 * gscale;
  if (l1 != 0.f) g += l1 * (w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f));
  g += wdv * w;
  mi = b1 * mi + (1.f - b1) * g;
  vi = b2 * vi + (1.f - b2) * g * g;
  return w - lr_step * mi * hw_rcp(hw_sqrt(vi) * inv_sqrt_bc2 + eps);
}
// 16 bytes per lane and array (the arenas are 16-byte aligned; the last n % 4 elements go through the scalar tail): the launch is a
// handful of microseconds of pure latency at these sizes (0.2-0.9 M elements), so fewer, wider requests per lane are what shortens it
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ wd, int64_t n, float lr,
                                                   float b1, float b2, float eps, float gscale, float l1,
                                                   const int32_t* __restrict__ step, unsigned short* __restrict__ p_hi,
                                                   unsigned short* __restrict__ p_lo, float* __restrict__ abs_partial, int clear) {
  __shared__ float red[4];
  float asum = 0.f;
  const int t = *step + 1;   // the launcher bumps *step after this kernel (a last-workgroup-bumps-it form was measured: the
                             // 2048 arrivals on one counter cost 15 us, the second launch 4)
  const float bc1 = 1.f - hw_exp2((float)t * hw_log2(b1));      // 1 - b1^t
  const float bc2 = 1.f - hw_exp2((float)t * hw_log2(b2));
  const float step_size = lr * hw_rcp(bc1);
  const float inv_sqrt_bc2 = hw_rsq(bc2);
  const int64_t n4 = n >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const float4 w4 = reinterpret_cast<const float4*>(p)[q], g4 = reinterpret_cast<const float4*>(grad)[q];
    float4 m4 = reinterpret_cast<const float4*>(m)[q], v4 = reinterpret_cast<const float4*>(v)[q];
    const float4 d4 = wd ? reinterpret_cast<const float4*>(wd)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (clear) reinterpret_cast<float4*>(grad)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    asum += (fabsf(w4.x) + fabsf(w4.y)) + (fabsf(w4.z) + fabsf(w4.w));
    float4 o;
    o.x = adam_elem(w4.x, g4.x, m4.x, v4.x, d4.x, step_size, b1, b2, eps, gscale, l1, inv_sqrt_bc2);
    o.y = adam_elem(w4.y, g4.y, m4.y, v4.y, d4.y, step_size, b1, b2, eps, gscale, l1, inv_sqrt_bc2);
    o.z = adam_elem(w4.z, g4.z, m4.z, v4.z, d4.z, step_size, b1, b2, eps, gscale, l1, inv_sqrt_bc2);
    o.w = adam_elem(w4.w, g4.w, m4.w, v4.w, d4.w, step_size, b1, b2, eps, gscale, l1, inv_sqrt_bc2);
    reinterpret_cast<float4*>(m)[q] = m4;
    reinterpret_cast<float4*>(v)[q] = v4;
    reinterpret_cast<float4*>(p)[q] = o;
    if (p_hi) {   // bf16x3 operand planes of the updated weights (hi = bf16(w), lo = bf16(w - hi)): the contractions read these
      const float ov[4] = {o.x, o.y, o.z, o.w};
      union { __bf16 b[4]; uint2 u; } hh, ll;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hh.b[j] = (__bf16)ov[j];
        ll.b[j] = (__bf16)(ov[j] - (float)hh.b[j]);
      }
      reinterpret_cast<uint2*>(p_hi)[q] = hh.u;
      reinterpret_cast<uint2*>(p_lo)[q] = ll.u;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // scalar tail
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float w = p[i];
    float mi = m[i], vi = v[i];
    const float wn = adam_elem(w, grad[i], mi, vi, wd ? wd[i] : 0.f, step_size, b1, b2, eps, gscale, l1, inv_sqrt_bc2);
    if (clear) grad[i] = 0.f;
    asum += fabsf(w);
    m[i] = mi; v[i] = vi; p[i] = wn;
    if (p_hi) {
      const __bf16 h = (__bf16)wn;
      const __bf16 l = (__bf16)(wn - (float)h);
      p_hi[i] = *reinterpret_cast<const unsigned short*>(&h);
      p_lo[i] = *reinterpret_cast<const unsigned short*>(&l);
    }
  }
  if (abs_partial) {
    asum = wave_sum(asum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = asum;
    __syncthreads();
    if (threadIdx.x == 0) abs_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}
__global__ void step_inc_kernel(int32_t* step) { *step += 1; }
// the step counter's increment and the RNG seed's advance of an optimizer step's end as ONE one-thread launch
__global__ void step_seed_tick_kernel(int32_t* step, int32_t* step2, uint64_t* seed, uint64_t inc) {
  if (step) *step += 1;
  if (step2) *step2 += 1;
  if (seed) *seed += inc;
}

extern "C" int advmil_adam_blocks(int64_t n) {
  if (n <= 0) return 0;
  const int64_t b = ((n >> 2) + 255) / 256;          // one lane = four elements
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
extern "C" int advmil_adam_step(float* p, float* grad, float* m, float* v, const float* wd, int64_t n, float lr,
                                float beta1, float beta2, float eps, float grad_scale, float l1_coef, int32_t* step,
                                void* p_hi, void* p_lo, int tick, float* abs_partial, int clear_grad, advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!p || !grad || !m || !v || !step || n <= 0 || ((p_hi != nullptr) != (p_lo != nullptr))) return ADVMIL_EINVAL;
  // the arenas are walked in 16-byte units (their planes in 8-byte units)
  if ((((uintptr_t)p | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v | (uintptr_t)wd) & 15) || (((uintptr_t)p_hi | (uintptr_t)p_lo) & 7)) return ADVMIL_EINVAL;
  const int blocks = advmil_adam_blocks(n);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, grad, m, v, wd, n, lr, beta1, beta2, eps, grad_scale,
                     l1_coef, step, (unsigned short*)p_hi, (unsigned short*)p_lo, abs_partial, clear_grad);
  if (tick) hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, stream, step);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_step_seed_tick(int32_t* step, int32_t* step2, uint64_t* seed, uint64_t inc, advmil_stream_t stream_) {
  if ((!step && !step2 && !seed) || (step && step == step2)) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(step_seed_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, step, step2, seed, inc);
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

// Bag ingest out of the device-resident cache (advmil_amd/ingest.py::SlabStager.add_device; replaces the per-epoch `.cuda()` of
// reference model/model_handler.py:315): up to three device-to-device range copies in ONE launch -- a cached bag's fp32 rows and its
// two operand planes into their places in the step slab. blockIdx.y = range; 16-byte units, four loads in flight per thread.
struct StageSpan { uint4* dst; const uint4* src; int64_t n16; };
__global__ __launch_bounds__(256) void stage_bag_kernel(StageSpan a, StageSpan b, StageSpan c) {
  const StageSpan sp = blockIdx.y == 0 ? a : (blockIdx.y == 1 ? b : c);
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < sp.n16; i += 4 * stride) {
    const uint4 v0 = sp.src[i], v1 = sp.src[i + stride], v2 = sp.src[i + 2 * stride], v3 = sp.src[i + 3 * stride];
    sp.dst[i] = v0; sp.dst[i + stride] = v1; sp.dst[i + 2 * stride] = v2; sp.dst[i + 3 * stride] = v3;
  }
  for (; i < sp.n16; i += stride) sp.dst[i] = sp.src[i];
}
// Same copy of the fp32 rows, with the two operand planes DERIVED on the way (hi = bf16(x), lo = bf16(x - hi): the rounding of
// advmil_split_planes and of the contraction kernels' own staging): the cache then keeps 4 bytes per element instead of 8 and a staged
// bag moves 12 bytes per element instead of 16. One thread = 8 consecutive floats (two 16-byte loads, two fp32 stores, one 16-byte
// store per plane).
__global__ __launch_bounds__(256) void stage_bag_split_kernel(const float* src, float* dst, int64_t n8, uint4* __restrict__ hi,
                                                              uint4* __restrict__ lo) {
  const bool copy = dst != src;                        // (in place: rows that just arrived over PCIe only get their planes)
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n8; idx += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * idx], b = reinterpret_cast<const float4*>(src)[2 * idx + 1];
    if (copy) {
      reinterpret_cast<float4*>(dst)[2 * idx] = a;
      reinterpret_cast<float4*>(dst)[2 * idx + 1] = b;
    }
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    union { __bf16 v[8]; uint4 u; } h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h.v[j] = (__bf16)x[j];
      l.v[j] = (__bf16)(x[j] - (float)h.v[j]);
    }
    hi[idx] = h.u;
    lo[idx] = l.u;
  }
}

extern "C" int advmil_stage_bag(void* dst_rows, const void* src_rows, size_t rows_bytes, void* dst_hi, const void* src_hi, void* dst_lo,
                                const void* src_lo, size_t plane_bytes, advmil_stream_t stream_) {
  if (!dst_rows || !src_rows || rows_bytes == 0 || (rows_bytes & 15)) return ADVMIL_EINVAL;
  const bool planes = dst_hi || src_hi || dst_lo || src_lo || plane_bytes;
  const bool split = planes && !src_hi && !src_lo;                 // planes wanted, none held: derive them from the rows
  if (planes && (!dst_hi || !dst_lo || plane_bytes == 0 || (plane_bytes & 15))) return ADVMIL_EINVAL;
  if (planes && !split && (!src_hi || !src_lo)) return ADVMIL_EINVAL;
  if (((uintptr_t)dst_rows | (uintptr_t)src_rows | (uintptr_t)dst_hi | (uintptr_t)src_hi | (uintptr_t)dst_lo | (uintptr_t)src_lo) & 15)
    return ADVMIL_EINVAL;
  if (split) {
    if (plane_bytes * 2 != rows_bytes || (rows_bytes & 31)) return ADVMIL_EINVAL;      // bf16 planes of fp32 rows, 8 floats per thread
    const int64_t n8 = (int64_t)(rows_bytes >> 5);
    // ADVMIL_STAGE_BLOCKS: workgroups of a staging launch. The launch runs on the copy stream UNDER the step's kernels, whose persistent
    // contractions hold one workgroup per CU and are bound by that CU's load path: a grid that floods every CU (4096 blocks until round 5)
    // slows all of them, 192 blocks leave most CUs alone at a time (resident product loop 3.17 -> 3.12 ms per step; 64: 3.34, 128: 3.16,
    // 256: 3.24; tools/probe/exp_staging.sh)
    static const int64_t cap = []() { const char* e = getenv("ADVMIL_STAGE_BLOCKS"); return (int64_t)(e ? atoi(e) : 192); }();
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(stage_bag_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const float*)src_rows,
                       (float*)dst_rows, n8, (uint4*)dst_hi, (uint4*)dst_lo);
    ADVMIL_LAUNCH_CHECK();
    return ADVMIL_OK;
  }
  StageSpan a{(uint4*)dst_rows, (const uint4*)src_rows, (int64_t)(rows_bytes >> 4)};
  StageSpan b{(uint4*)dst_hi, (const uint4*)src_hi, planes ? (int64_t)(plane_bytes >> 4) : 0};
  StageSpan c{(uint4*)dst_lo, (const uint4*)src_lo, planes ? (int64_t)(plane_bytes >> 4) : 0};
  static const int64_t cap3 = []() { const char* e = getenv("ADVMIL_STAGE_BLOCKS"); return (int64_t)(e ? atoi(e) : 192); }();
  int64_t blocks = ((int64_t)(rows_bytes >> 4) + 1023) / 1024;
  if (blocks > cap3) blocks = cap3;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(stage_bag_kernel, dim3((unsigned)blocks, planes ? 3 : 1), dim3(256), 0, (hipStream_t)stream_, a, b, c);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// rng_row (optional, with the row width `width`): element i of the flat tensor belongs to row i / width, whose draws are indexed
// as row rng_row[i / width] -- the row it would occupy in the single-process run (bag-parallel world-size invariance)
__device__ __forceinline__ uint64_t rng_flat_index(int64_t i, const int64_t* rng_row, int64_t width) {
  return rng_row ? (uint64_t)(rng_row[i / width] * width + i % width) : (uint64_t)i;
}
__global__ __launch_bounds__(256) void uniform_fill_kernel(float* __restrict__ out, int64_t n, const uint64_t* seed, uint64_t stream_id,
                                                           const int64_t* __restrict__ rng_row, int64_t width) {
  const uint64_t key = rng_key(*seed, stream_id);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = rng_uniform(key, rng_flat_index(i, rng_row, width));
}
extern "C" int advmil_uniform_fill(float* out, int64_t n, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row,
                                   int64_t width, advmil_stream_t stream_) {
  if (!out || !seed || n <= 0 || (rng_row && width <= 0)) return ADVMIL_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(uniform_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, out, n, seed, stream_id, rng_row, width);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// y[i] = x[i] * (u(seed, stream, i) >= p ? 1/(1-p) : 0): the dropout of the [B, d]-sized head tensors (and, applied to dy, its
// backward) as ONE launch; same draw as the GEMM epilogue's dropout and synth.dropout_keep.
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p,
                                                            const uint64_t* seed, uint64_t stream_id,
                                                            const int64_t* __restrict__ rng_row, int64_t width) {
  const uint64_t key = rng_key(*seed, stream_id);
  const float inv = hw_rcp(1.0f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * rng_keep(key, rng_flat_index(i, rng_row, width), p, inv);
}
extern "C" int advmil_dropout_apply(const float* x, float* y, int64_t n, float p, const uint64_t* seed, uint64_t stream_id,
                                    const int64_t* rng_row, int64_t width, advmil_stream_t stream_) {
  if (!x || !y || !seed || n <= 0 || !(p >= 0.0f && p < 1.0f) || (rng_row && width <= 0)) return ADVMIL_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, y, n, p, seed, stream_id, rng_row, width);
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

// Device wall-clock stamp (wall_clock64: the constant-rate counter, advmil_clock_rate_khz ticks per ms): a one-thread launch the
// measurement side (bench.py) puts before and after a kernel INSIDE the captured step graph, so a launch is timed where it runs in the
// step -- between its real neighbours, at the clock the step's power draw allows -- not in a back-to-back replay.
__global__ void stamp_clock_kernel(int64_t* dst) { *dst = (int64_t)wall_clock64(); }
extern "C" int advmil_stamp_clock(int64_t* dst, advmil_stream_t stream_) {
  if (!dst) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(stamp_clock_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, dst);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
extern "C" int64_t advmil_clock_rate_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return (int64_t)khz;
}

// =====================================================================================
// The two scalar losses of the step over <= bp_every_batch values, value AND analytic gradient in one launch each
// (the reference composes ~25 elementwise/reduce ops per loss and autograd as many again: loss/utils.py:21-41, 182-208).
// =====================================================================================
// which: 0 = bce as shipped (fake term -(1 - log(sigmoid(f) + 1e-8)), real term -log(sigmoid(r) + 1e-8)), 1 = hinge, 2 = wasserstein.
// loss = inv_nf * sum_i term_f(fake_i) + inv_nr * sum_i mask_i * term_r(real_i); out3 = {loss, sum_i mask_i*real_i, sum_i fake_i}
__global__ __launch_bounds__(256) void gan_d_loss_kernel(const float* __restrict__ fake, int nf, const float* __restrict__ real,
                                                         const float* __restrict__ mask, int nr, int which, float inv_nf,
                                                         float inv_nr, float* __restrict__ out3, float* __restrict__ g_fake,
                                                         float* __restrict__ g_real) {
  float lf = 0.f, sf = 0.f, lr = 0.f, sr = 0.f;
  // (selects, not a three-way branch: hipcc 7.2 mis-structured the branchy form -- on the wasserstein path the assignment
  //  term = -r was hoisted into a block that path never executes; <= bp_every_batch elements, so evaluating all forms is free)
  for (int i = threadIdx.x; i < nf; i += 256) {
    const float f = fake[i];
    const float sg = hw_rcp(1.0f + hw_exp(-f));
    const float t_bce = -(1.0f - hw_log(sg + 1e-8f)), d_bce = sg * (1.0f - sg) * hw_rcp(sg + 1e-8f);
    const float t_hin = fmaxf(1.0f + f, 0.0f), d_hin = (1.0f + f > 0.0f) ? 1.0f : 0.0f;
    const float term = which == 0 ? t_bce : (which == 1 ? t_hin : f);
    const float d = which == 0 ? d_bce : (which == 1 ? d_hin : 1.0f);
    lf += term; sf += f;
    g_fake[i] = d * inv_nf;
  }
  for (int i = threadIdx.x; i < nr; i += 256) {
    const float r = real[i], m = mask ? mask[i] : 1.0f;
    const float sg = hw_rcp(1.0f + hw_exp(-r));
    const float t_bce = -hw_log(sg + 1e-8f), d_bce = -sg * (1.0f - sg) * hw_rcp(sg + 1e-8f);
    const float t_hin = fmaxf(1.0f - r, 0.0f), d_hin = (1.0f - r > 0.0f) ? -1.0f : 0.0f;
    const float term = which == 0 ? t_bce : (which == 1 ? t_hin : -r);
    const float d = which == 0 ? d_bce : (which == 1 ? d_hin : -1.0f);
    lr += m * term; sr += m * r;
    g_real[i] = m * d * inv_nr;
  }
  // one barrier: every wave leaves its four partial sums in LDS, thread 0 adds the 4 x 4 values
  __shared__ float red4[4][4];
  lf = wave_sum(lf); lr = wave_sum(lr); sr = wave_sum(sr); sf = wave_sum(sf);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red4[0][w] = lf; red4[1][w] = lr; red4[2][w] = sr; red4[3][w] = sf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (red4[q][0] + red4[q][1]) + (red4[q][2] + red4[q][3]);
    out3[0] = t[0] * inv_nf + t[1] * inv_nr;
    out3[1] = t[2];
    out3[2] = t[3];
  }
}

extern "C" int advmil_gan_d_loss(const float* fake, int nf, const float* real, const float* real_mask, int nr, int which,
                                 float inv_nf, float inv_nr, float* out3, float* g_fake, float* g_real, advmil_stream_t stream_) {
  if (!fake || !out3 || !g_fake || nf <= 0 || nr < 0 || (nr > 0 && (!real || !g_real)) || which < 0 || which > 2) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(gan_d_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, fake, nf, real, real_mask, nr, which, inv_nf,
                     inv_nr, out3, g_fake, g_real);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// Generator loss (model_handler.py:468-486): gen = -inv_nf * sum_i fake_i; reg = inv_nv * sum_i vis_i * recon_term(pred_i, t_i, e_i)
// with recon_term = (1-alpha) * (obs + cen) + alpha * obs, obs = e|p - t|, cen = (1-e) relu(gamma - (p - t)), both squared for l2
// (loss/utils.py:21-41); total = reg + coef * gen. out3 = {total, reg, gen}; g_pred = d total / d pred, g_fake = d total / d fake.
__global__ __launch_bounds__(256) void gan_g_loss_kernel(const float* __restrict__ pred, const float* __restrict__ t,
                                                         const float* __restrict__ e, const float* __restrict__ vis,
                                                         const float* __restrict__ fake, int n, float alpha, float gamma, int l2,
                                                         float coef, float inv_nf, float inv_nv, float* __restrict__ out3,
                                                         float* __restrict__ g_pred, float* __restrict__ g_fake) {
  float sreg = 0.f, sfake = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float p = pred[i], ti = t[i], ei = e[i], v = vis ? vis[i] : 1.0f;
    const float df = p - ti;
    float obs = ei * fabsf(df), cen = (1.0f - ei) * fmaxf(gamma - df, 0.0f);
    float dobs = ei * (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f));
    float dcen = (gamma - df > 0.0f) ? -(1.0f - ei) : 0.0f;
    if (l2) { dobs = 2.0f * obs * dobs; dcen = 2.0f * cen * dcen; obs *= obs; cen *= cen; }
    sreg += v * ((1.0f - alpha) * (obs + cen) + alpha * obs);
    g_pred[i] = inv_nv > 0.0f ? v * inv_nv * ((1.0f - alpha) * (dobs + dcen) + alpha * dobs) : 0.0f;
    sfake += fake[i];
    g_fake[i] = -coef * inv_nf;
  }
  __shared__ float red2[2][4];
  sreg = wave_sum(sreg); sfake = wave_sum(sfake);
  if ((threadIdx.x & 63) == 0) { red2[0][threadIdx.x >> 6] = sreg; red2[1][threadIdx.x >> 6] = sfake; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sreg = (red2[0][0] + red2[0][1]) + (red2[0][2] + red2[0][3]);
    sfake = (red2[1][0] + red2[1][1]) + (red2[1][2] + red2[1][3]);
    const float reg = sreg * inv_nv, gen = -sfake * inv_nf;
    out3[0] = reg + coef * gen; out3[1] = reg; out3[2] = gen;
  }
}

extern "C" int advmil_gan_g_loss(const float* pred, const float* t, const float* e, const float* vis_mask, const float* fake, int n,
                                 float alpha, float gamma, int l2, float coef, float inv_nf, float inv_nv, float* out3, float* g_pred,
                                 float* g_fake, advmil_stream_t stream_) {
  if (!pred || !t || !e || !fake || !out3 || !g_pred || !g_fake || n <= 0) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(gan_g_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, pred, t, e, vis_mask, fake, n, alpha, gamma, l2,
                     coef, inv_nf, inv_nv, out3, g_pred, g_fake);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// =====================================================================================
// Skinny Linear layers of the heads (in_features == 1: the first layer of the y-embedding, model_utils.py:178-186; out_features
// == 1: the projection layer GANSurv.py:78-84 and the generator's output layer): y[b,n] = act(sum_k x[b,k] W[n,k] + bias[n]) for
// B <= a step's bags. One launch forward; one launch backward producing dx, dW, dbias (accumulated into the caller's buffers).
// =====================================================================================
__global__ __launch_bounds__(256) void skinny_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, int B, int K, int N, int act,
                                                                float* __restrict__ y) {
  if (N == 1) {             // one wave per row: lanes stride over k (a single thread walking K = 128..768 took 12-14 us)
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += gridDim.x * 4) {
      float s = 0.0f;
      for (int k = lane; k < K; k += 64) s += x[(int64_t)b * K + k] * W[k];
      s = wave_sum(s);
      if (lane == 0) y[b] = act_apply(act, s + (bias ? bias[0] : 0.0f));
    }
    return;
  }
  for (int o = blockIdx.x * 256 + threadIdx.x; o < B * N; o += gridDim.x * 256) {
    const int b = o / N, n = o % N;
    float s = bias ? bias[n] : 0.0f;
    for (int k = 0; k < K; ++k) s += x[(int64_t)b * K + k] * W[(int64_t)n * K + k];
    y[o] = act_apply(act, s);
  }
}

// outputs indexed 0..N*K-1: dW; N*K..N*K+N-1: dbias; then B*K: dx
__global__ __launch_bounds__(256) void skinny_linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ y, const float* __restrict__ dy, int B,
                                                                int K, int N, int act, float* __restrict__ dx,
                                                                float* __restrict__ dW, float* __restrict__ dbias, int accumulate) {
  const int nW = dW ? N * K : 0, nb = dbias ? N : 0, nx = dx ? B * K : 0;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < nW + nb + nx; o += gridDim.x * 256) {
    if (o < nW) {
      const int n = o / K, k = o % K;
      float s = 0.0f;
      for (int b = 0; b < B; ++b) s += dy[b * N + n] * act_grad_from_out(act, y[b * N + n]) * x[(int64_t)b * K + k];
      dW[o] = accumulate ? dW[o] + s : s;
    } else if (o < nW + nb) {
      const int n = o - nW;
      float s = 0.0f;
      for (int b = 0; b < B; ++b) s += dy[b * N + n] * act_grad_from_out(act, y[b * N + n]);
      dbias[n] = accumulate ? dbias[n] + s : s;
    } else {
      const int q = o - nW - nb, b = q / K, k = q % K;
      float s = 0.0f;
      for (int n = 0; n < N; ++n) s += dy[b * N + n] * act_grad_from_out(act, y[b * N + n]) * W[(int64_t)n * K + k];
      dx[q] = s;
    }
  }
}

// =====================================================================================
// Linear layers on [B, d] head / tail tensors (B = the bags of one optimizer step, <= 32 rows): hop MLP and rho of the generator,
// the discriminator's bag-level MLPs and label embedding (reference model/GANSurv.py:30-49, 89-105; model_utils.py:116-186).
// Round 3 ran them on the 64x64-tile MFMA contraction: 7-10 us per launch in-graph (its K walk is a serial chain of dependent MFMAs
// on ONE accumulator block: 0.44 us per 32 k whatever the prefetch depth), and a backward was up to six launches (activation /
// dropout backward, dW contraction + split-K reduce, dX contraction, bias column sum + merge). Here: plain fp32 FMA, one wave per 4
// output columns with the 64 lanes splitting K (1 KB contiguous weight reads), a value-halving butterfly for the lane reduction
// (V values cost ~V shuffles instead of 6 V), bias / activation / dropout in the same launch; backward = ONE launch for dpre, dW,
// dbias (+ one for dX when the input needs a gradient). Same dropout draw as the contraction epilogue (rng_keep(key, row * N + col)).
// =====================================================================================
// sum of v[i] over the 64 lanes for V (a power of two) values per lane: stage d keeps half of the values and trades the other half
// with lane ^ d. Afterwards lane l holds the totals of the original indices base(l) + [0, max(V / 64, 1)) with
// base(l) = sum over the first log2(V) stages s of ((l >> (5 - s)) & 1) * (V >> (s + 1)); written to out[V] (wave-private LDS).
template <int V>
__device__ __forceinline__ void wave_multi_reduce(float (&v)[V], int lane, float* __restrict__ out) {
  int base = 0;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int d = 32 >> s;
    const int len = (V >> s) > 1 ? (V >> s) : 1;       // values per lane entering the stage
    if (len > 1) {
      const int half = len / 2;
      const bool up = (lane & d) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const float a = v[i], b = v[i + half];
        const float send = up ? a : b, keep = up ? b : a;
        v[i] = keep + __shfl_xor(send, d, 64);
      }
      if (up) base += half;
    } else {
      v[0] += __shfl_xor(v[0], d, 64);
    }
  }
  constexpr int R = V >= 64 ? V / 64 : 1;
#pragma unroll
  for (int i = 0; i < R; ++i) out[base + i] = v[i];
}

// MT = rows of the tile (>= M, power of two). grid = ceil(N / 16) blocks of 4 waves; wave w of block g owns columns 16 g + 4 w .. + 3
template <int MT>
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W,
                                                               const float* __restrict__ bias, int M, int N, int K, int act, float p,
                                                               const uint64_t* seed, uint64_t stream_id, const int64_t* __restrict__ rng_row,
                                                               float* __restrict__ y) {
  __shared__ float red[4][4 * MT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = ((int)blockIdx.x * 4 + wave) * 4;
  if (n0 >= N) return;
  float acc[4 * MT];
#pragma unroll
  for (int i = 0; i < 4 * MT; ++i) acc[i] = 0.f;
  for (int k = 4 * lane; k < K; k += 256) {
    float4 wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wv[j] = n0 + j < N ? *reinterpret_cast<const float4*>(W + (int64_t)(n0 + j) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j * MT + m] += (xv.x * wv[j].x + xv.y * wv[j].y) + (xv.z * wv[j].z + xv.w * wv[j].w);
      }
    }
  }
  wave_multi_reduce<4 * MT>(acc, lane, red[wave]);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < 4 * MT) {
    const int j = lane / MT, m = lane % MT, n = n0 + j;
    if (m < M && n < N) {
      float v = red[wave][lane] + (bias ? bias[n] : 0.f);
      v = act_apply(act, v);
      if (seed && p > 0.f) {
        const int64_t grow = rng_row ? rng_row[m] : m;
        v *= rng_keep(rng_key(*seed, stream_id), (uint64_t)(grow * N + n), p, hw_rcp(1.f - p));
      }
      y[(int64_t)m * N + n] = v;
    }
  }
  if (4 * MT > 64 && lane + 64 < 4 * MT) {                       // (MT = 32: 128 results per wave)
    const int t = lane + 64, j = t / MT, m = t % MT, n = n0 + j;
    if (m < M && n < N) {
      float v = red[wave][t] + (bias ? bias[n] : 0.f);
      v = act_apply(act, v);
      if (seed && p > 0.f) {
        const int64_t grow = rng_row ? rng_row[m] : m;
        v *= rng_keep(rng_key(*seed, stream_id), (uint64_t)(grow * N + n), p, hw_rcp(1.f - p));
      }
      y[(int64_t)m * N + n] = v;
    }
  }
}

// backward, weights side: dpre[m, n] = dy * keep * act'(y) (also written to `dpre_out` for the dX launch), dW[n, :] (+)= sum_m dpre x[m, :],
// dbias[n] (+)= sum_m dpre. Same grid as the forward.
template <int MT>
__global__ __launch_bounds__(256) void small_linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ x, int64_t ldx, int M, int N, int K, int act, float p,
                                                                 const uint64_t* seed, uint64_t stream_id, const int64_t* __restrict__ rng_row,
                                                                 float* __restrict__ dW, int acc_w, float* __restrict__ dbias, int acc_b,
                                                                 float* __restrict__ dpre_out) {
  __shared__ float dps[4][4 * MT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = ((int)blockIdx.x * 4 + wave) * 4;
  if (n0 >= N) return;
  const bool drop = seed && p > 0.f;
  uint64_t key = 0;
  float inv = 1.f;
  if (drop) { key = rng_key(*seed, stream_id); inv = hw_rcp(1.f - p); }
#pragma unroll
  for (int t0 = 0; t0 < 4 * MT; t0 += 64) {
    const int t = t0 + lane;
    if (t < 4 * MT) {
      const int j = t / MT, m = t % MT, n = n0 + j;
      float d = 0.f;
      if (m < M && n < N) {
        float f = 1.f, yy = y[(int64_t)m * N + n];
        if (drop) {
          const int64_t grow = rng_row ? rng_row[m] : m;
          f = rng_keep(key, (uint64_t)(grow * N + n), p, inv);
          yy *= 1.f - p;                       // undo the 1/(1-p) on kept elements (dropped ones get f = 0 anyway)
        }
        d = dy[(int64_t)m * N + n] * f * act_grad_from_out(act, yy);
        if (dpre_out) dpre_out[(int64_t)m * N + n] = d;
      }
      dps[wave][t] = d;
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float dp[4 * MT];
#pragma unroll
  for (int i = 0; i < 4 * MT; ++i) dp[i] = dps[wave][i];
  if (dbias && lane < 4 && n0 + lane < N) {
    float sb = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) sb += dps[wave][lane * MT + m];
    if (acc_b) sb += dbias[n0 + lane];
    dbias[n0 + lane] = sb;
  }
  if (!dW) return;
  for (int k = 4 * lane; k < K; k += 256) {
    float4 g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = dp[j * MT + m];
          g[j].x += d * xv.x; g[j].y += d * xv.y; g[j].z += d * xv.z; g[j].w += d * xv.w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n0 + j < N) {
        float4* dst = reinterpret_cast<float4*>(dW + (int64_t)(n0 + j) * K + k);
        if (acc_w) { const float4 o = *dst; g[j].x += o.x; g[j].y += o.y; g[j].z += o.z; g[j].w += o.w; }
        *dst = g[j];
      }
    }
  }
}

// backward, input side: dx[m, k] = sum_n dpre[m, n] W[n, k]. Block = 8 waves splitting n, lane = 4 consecutive k; rows in chunks of 8:
// the chunk's dpre rows are staged in LDS first (a wave-uniform global load inside the n loop made every step wait a full memory
// latency: 48 dependent steps = 14 us on a [16, 384] layer), the weight rows are fetched four at a time; the cross-wave sums go
// through LDS (8 waves x 8 rows x 256 k floats = 64 KB). N <= SMALL_NMAX.
#define SMALL_NMAX 1024
__global__ __launch_bounds__(512) void small_linear_bwd_x_kernel(const float* __restrict__ dpre, const float* __restrict__ W, int M, int N,
                                                                 int K, float* __restrict__ dx, int64_t lddx) {
  __shared__ float4 part[8][8][64];
  __shared__ __attribute__((aligned(16))) float dps[8][SMALL_NMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = ((int)blockIdx.x * 64 + lane) * 4;
  const bool kok = k < K;
  const int nper = (((N + 7) / 8) + 3) & ~3;                     // columns per wave, a multiple of 4
  const int nb = wave * nper, ne = nb + nper < N ? nb + nper : N;
  for (int m0 = 0; m0 < M; m0 += 8) {
    for (int e = tid; e < 8 * N; e += 512) {
      const int i = e / N, n = e - i * N;
      dps[i][n] = m0 + i < M ? dpre[(int64_t)(m0 + i) * N + n] : 0.f;
    }
    __syncthreads();
    float4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kok) {
      int n = nb;
      for (; n + 4 <= ne; n += 4) {
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[u] = *reinterpret_cast<const float4*>(W + (int64_t)(n + u) * K + k);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 d4 = *reinterpret_cast<const float4*>(&dps[i][n]);        // (n is a multiple of 4: nb, nper are)
          const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc[i].x += dv[u] * wv[u].x; acc[i].y += dv[u] * wv[u].y; acc[i].z += dv[u] * wv[u].z; acc[i].w += dv[u] * wv[u].w;
          }
        }
      }
      for (; n < ne; ++n) {
        const float4 wv = *reinterpret_cast<const float4*>(W + (int64_t)n * K + k);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = dps[i][n];
          acc[i].x += d * wv.x; acc[i].y += d * wv.y; acc[i].z += d * wv.z; acc[i].w += d * wv.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[wave][i][lane] = acc[i];
    __syncthreads();
    {
      const int i = wave;                          // wave i sums row m0 + i over the 8 partials
      if (m0 + i < M && kok) {
        float4 s4 = part[0][i][lane];
#pragma unroll
        for (int w2 = 1; w2 < 8; ++w2) { const float4 q = part[w2][i][lane]; s4.x += q.x; s4.y += q.y; s4.z += q.z; s4.w += q.w; }
        *reinterpret_cast<float4*>(dx + (int64_t)(m0 + i) * lddx + k) = s4;
      }
    }
    __syncthreads();
  }
}

static int small_mt(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : M <= 8 ? 8 : M <= 16 ? 16 : 32; }
#define SMALL_DISPATCH(KERNEL, MT_, GRID, STREAM, ...)                                                            \
  do {                                                                                                            \
    switch (MT_) {                                                                                                \
      case 1: hipLaunchKernelGGL((KERNEL<1>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); break;                    \
      case 2: hipLaunchKernelGGL((KERNEL<2>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); break;                    \
      case 4: hipLaunchKernelGGL((KERNEL<4>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); break;                    \
      case 8: hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); break;                    \
      case 16: hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); break;                  \
      default: hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); break;                  \
    }                                                                                                             \
  } while (0)

extern "C" int advmil_small_linear_fwd(const float* x, int64_t ldx, const float* W, const float* bias, int M, int N, int K, int act,
                                       float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, float* y,
                                       advmil_stream_t stream_) {
  if (!x || !W || !y || M <= 0 || M > 32 || N <= 0 || K <= 0 || (K & 3) || (ldx & 3) || ldx < K) return ADVMIL_EINVAL;
  if (((uintptr_t)x | (uintptr_t)W) & 15) return ADVMIL_EINVAL;
  if (!(drop_p >= 0.f && drop_p < 1.f)) return ADVMIL_EINVAL;
  const dim3 grid((unsigned)((N + 15) / 16));
  SMALL_DISPATCH(small_linear_fwd_kernel, small_mt(M), grid, (hipStream_t)stream_, x, ldx, W, bias, M, N, K, act, drop_p, seed, stream_id,
                 rng_row, y);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" size_t advmil_small_linear_bwd_workspace_bytes(int M, int N) { return (size_t)M * (size_t)N * sizeof(float); }

extern "C" int advmil_small_linear_bwd(const float* dy, const float* y, const float* x, int64_t ldx, const float* W, int M, int N, int K,
                                       int act, float drop_p, const uint64_t* seed, uint64_t stream_id, const int64_t* rng_row, float* dW,
                                       int acc_w, float* dbias, int acc_b, float* dx, int64_t lddx, void* ws, size_t ws_bytes,
                                       advmil_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dy || !y || !x || !W || M <= 0 || M > 32 || N <= 0 || K <= 0 || (K & 3) || (ldx & 3) || ldx < K) return ADVMIL_EINVAL;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)dW | (uintptr_t)dx) & 15) return ADVMIL_EINVAL;
  if (dx && ((lddx & 3) || lddx < K || !ws || ws_bytes < advmil_small_linear_bwd_workspace_bytes(M, N) || N > SMALL_NMAX)) return ADVMIL_EINVAL;
  if (!(drop_p >= 0.f && drop_p < 1.f)) return ADVMIL_EINVAL;
  const dim3 grid((unsigned)((N + 15) / 16));
  SMALL_DISPATCH(small_linear_bwd_w_kernel, small_mt(M), grid, stream, dy, y, x, ldx, M, N, K, act, drop_p, seed, stream_id, rng_row, dW,
                 acc_w, dbias, acc_b, dx ? (float*)ws : (float*)nullptr);
  ADVMIL_LAUNCH_CHECK();
  if (dx) {
    hipLaunchKernelGGL(small_linear_bwd_x_kernel, dim3((unsigned)((K + 255) / 256)), dim3(512), 0, stream, (const float*)ws, W, M, N, K, dx,
                       lddx);
    ADVMIL_LAUNCH_CHECK();
  }
  return ADVMIL_OK;
}

extern "C" int advmil_skinny_linear_fwd(const float* x, const float* W, const float* bias, int B, int K, int N, int act, float* y,
                                        advmil_stream_t stream_) {
  if (!x || !W || !y || B <= 0 || K <= 0 || N <= 0 || (int64_t)B * N > (1 << 20) || (K != 1 && N != 1)) return ADVMIL_EINVAL;
  const int blocks = N == 1 ? (B + 3) / 4 : (B * N + 255) / 256;
  hipLaunchKernelGGL(skinny_linear_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, x, W, bias, B, K, N, act, y);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

extern "C" int advmil_skinny_linear_bwd(const float* x, const float* W, const float* y, const float* dy, int B, int K, int N, int act,
                                        float* dx, float* dW, float* dbias, int accumulate, advmil_stream_t stream_) {
  if (!x || !W || !y || !dy || B <= 0 || K <= 0 || N <= 0 || (K != 1 && N != 1)) return ADVMIL_EINVAL;
  const int total = (dW ? N * K : 0) + (dbias ? N : 0) + (dx ? B * K : 0);
  if (total == 0) return ADVMIL_OK;
  hipLaunchKernelGGL(skinny_linear_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, x, W, y, dy, B, K, N, act,
                     dx, dW, dbias, accumulate);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}

// ---- projection head of the discriminator (GANSurv.py:78-105): out[b] = <u[b], t[b]> + <src[b], w> + bias -- the (region-level)
// inner product with the label embedding plus the width-1 projection layer, [B, d] head tensors: one launch each way instead of
// mul + sum + linear + add (and their four backward launches). One wave per row.
__global__ __launch_bounds__(64) void prj_head_fwd_kernel(const float* __restrict__ u, const float* __restrict__ t,
                                                          const float* __restrict__ src, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int d, float* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) {
    s += u[(int64_t)b * d + j] * t[(int64_t)b * d + j];
    if (src) s += src[(int64_t)b * d + j] * w[j];
  }
  s = wave_sum(s);
  if (lane == 0) out[b] = s + ((src && bias) ? bias[0] : 0.f);
}
// blocks 0..B-1: du[b] = dout[b] t[b], dt[b] = dout[b] u[b], dsrc[b] = dout[b] w; block B: dw[j] = sum_b dout[b] src[b][j], dbias = sum_b dout[b]
// (rows in index order: deterministic). accumulate != 0 adds into dw / dbias (the optimizer's gradient arena).
__global__ __launch_bounds__(256) void prj_head_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ u,
                                                           const float* __restrict__ t, const float* __restrict__ src,
                                                           const float* __restrict__ w, int B, int d, float* __restrict__ du,
                                                           float* __restrict__ dt, float* __restrict__ dsrc, float* __restrict__ dw,
                                                           float* __restrict__ dbias, int accumulate) {
  const int b = blockIdx.x;
  if (b < B) {
    const float g = dout[b];
    for (int j = threadIdx.x; j < d; j += 256) {
      const int64_t o = (int64_t)b * d + j;
      if (du) du[o] = g * t[o];
      if (dt) dt[o] = g * u[o];
      if (dsrc) dsrc[o] = g * w[j];
    }
    return;
  }
  if (dw)
    for (int j = threadIdx.x; j < d; j += 256) {
      float s = 0.f;
      for (int r = 0; r < B; ++r) s += dout[r] * src[(int64_t)r * d + j];
      dw[j] = accumulate ? dw[j] + s : s;
    }
  if (dbias && threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < B; ++r) s += dout[r];
    dbias[0] = accumulate ? dbias[0] + s : s;
  }
}
extern "C" int advmil_prj_head_fwd(const float* u, const float* t, const float* src, const float* w, const float* bias, int B, int d,
                                   float* out, advmil_stream_t stream_) {
  if (!u || !t || !out || B <= 0 || d <= 0 || ((src != nullptr) != (w != nullptr))) return ADVMIL_EINVAL;
  hipLaunchKernelGGL(prj_head_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream_, u, t, src, w, bias, d, out);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
extern "C" int advmil_prj_head_bwd(const float* dout, const float* u, const float* t, const float* src, const float* w, int B, int d,
                                   float* du, float* dt, float* dsrc, float* dw, float* dbias, int accumulate, advmil_stream_t stream_) {
  if (!dout || !u || !t || B <= 0 || d <= 0 || ((dsrc || dw) && (!src || !w))) return ADVMIL_EINVAL;
  const int extra = (dw || dbias) ? 1 : 0;
  hipLaunchKernelGGL(prj_head_bwd_kernel, dim3(B + extra), dim3(256), 0, (hipStream_t)stream_, dout, u, t, src, w, B, d, du, dt, dsrc,
                     dw, dbias, accumulate);
  ADVMIL_LAUNCH_CHECK();
  return ADVMIL_OK;
}
