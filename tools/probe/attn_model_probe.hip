// Dev probe: issue model of the attention forward's per-wave sequence on one gfx950 SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/attn_model_probe tools/probe/attn_model_probe.hip && tools/probe/attn_model_probe
// Every wave loops over "half-tiles": 9 MFMAs into one accumulator (QK^T) -> NV VALU that depend on it (softmax / dropout / split,
// modelled as v_fma chains over the 16 accumulator registers + 4 v_exp per 16) -> 12 MFMAs into two other accumulators whose B
// operand comes from that VALU (P.V). W waves per SIMD (one workgroup of 4 W waves per CU).
//   BAR   0 none, 1 workgroup barrier every 2 half-tiles
//   STAG  0 all waves start together, 1 wave w waits w/4 * (phase length / W) first
//   PRIO  0 none, 1 s_setprio 2 around the MFMA bursts, 2 static priority by wave (wave>>2)
// Output: cycles per half-tile per SIMD-wave (wave 0's elapsed / half-tiles) and the implied matrix-pipe utilisation
// ( W * 21 * 32 / cycles ).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int W, int BAR, int STAG, int PRIO, int NVX>
__global__ __launch_bounds__(256 * W) void model(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  f32x16 s, o0, o1;
  for (int r = 0; r < 16; ++r) { s[r] = 0.f; o0[r] = in[(r + lane) & 1023]; o1[r] = in[(r + 2 * lane) & 1023]; }
  union U { uint4 u; bf16x8 v; float f[4]; } fa, fb, p0, p1;
  fa.u = make_uint4(__float_as_uint(in[lane]), __float_as_uint(in[lane + 64]), __float_as_uint(in[lane + 128]), __float_as_uint(in[lane + 192]));
  fb.u = make_uint4(__float_as_uint(in[lane + 256]), __float_as_uint(in[lane + 320]), __float_as_uint(in[lane + 384]), __float_as_uint(in[lane + 448]));
  const float ca = 0.999f, cb = 0.001f;
  if (PRIO == 2 && (wave >> 2) == 0) __builtin_amdgcn_s_setprio(1);
  __syncthreads();
  if (STAG) for (int d = 0; d < (wave >> 2) * (1400 / W); d += 64) __builtin_amdgcn_s_sleep(1);
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, constant
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (PRIO == 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int k = 0; k < 9; ++k) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, s, 0, 0, 0);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    // VALU: NVX passes of (16 fma) + 4 exp + packing -> p0, p1
#pragma unroll
    for (int x = 0; x < NVX; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_fmaf(s[r], ca, cb);
    float e0 = s[0], e1 = s[5], e2 = s[10], e3 = s[15];
    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\ts_nop 1" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
    p0.f[0] = s[1] + e0; p0.f[1] = s[2] + e1; p0.f[2] = s[3] + e2; p0.f[3] = s[4] + e3;
    p1.f[0] = s[6] + e0; p1.f[1] = s[7] + e1; p1.f[2] = s[8] + e2; p1.f[3] = s[9] + e3;
    if (PRIO == 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, p0.v, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb.v, p1.v, o1, 0, 0, 0);
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = o0[r] * 1e-30f;      // next QK starts from something that depends on this P.V (keeps the chain honest)
    if (BAR && (it & 1)) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += s[r] + o0[r] + o1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int W, int BAR, int STAG, int PRIO, int NVX>
void run(const float* in, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  unsigned long long h, hr[2];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((model<W, BAR, STAG, PRIO, NVX>), dim3(256), dim3(256 * W), 0, 0, in, out, 100, cyc);      // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((model<W, BAR, STAG, PRIO, NVX>), dim3(256), dim3(256 * W), 0, 0, in, out, iters, cyc);
  hipEventRecord(e1, 0);
  hipError_t err = hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(hr, cyc, 16, hipMemcpyDeviceToHost);
  h = hr[0];
  const double ghz = (double)hr[0] / ((double)hr[1] * 10.0);          // s_memtime ticks per ns of s_memrealtime (100 MHz)
  const double per = (double)h / iters, ns_simd = 1e6 * ms / iters / W;      // wall ns per half-tile per SIMD (W waves share it)
  printf("W=%d bar=%d stag=%d prio=%d VALU=%3d+12 : wave0 %7.1f ticks per half-tile; wall %7.1f ns per half-tile per SIMD-wave = %.2f of the matrix-pipe time (21 MFMA x 32 cyc @ 2.4 GHz = 280 ns); wave 0: %.0f ns by s_memrealtime, s_memtime / s_memrealtime = %.2f ticks per ns%s\n",
         W, BAR, STAG, PRIO, 16 * NVX + 16, per, ns_simd, 280.0 / ns_simd, (double)hr[1] * 10.0 / iters, ghz, err == hipSuccess ? "" : "  LAUNCH ERROR");
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64);
  float hin[1024];
  for (int i = 0; i < 1024; ++i) hin[i] = 0.5f + 0.0001f * (float)((i * 2654435761u) % 1000);
  hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
  // ~230 VALU per half-tile = 14 passes of 16
  run<1, 0, 0, 0, 14>(in, out, cyc); run<2, 0, 0, 0, 14>(in, out, cyc); run<3, 0, 0, 0, 14>(in, out, cyc); run<4, 0, 0, 0, 14>(in, out, cyc);
  run<2, 1, 0, 0, 14>(in, out, cyc); run<4, 1, 0, 0, 14>(in, out, cyc);
  run<2, 0, 1, 0, 14>(in, out, cyc); run<4, 0, 1, 0, 14>(in, out, cyc); run<2, 1, 1, 0, 14>(in, out, cyc); run<4, 1, 1, 0, 14>(in, out, cyc);
  run<2, 0, 0, 1, 14>(in, out, cyc); run<4, 0, 0, 1, 14>(in, out, cyc); run<2, 1, 0, 1, 14>(in, out, cyc); run<4, 1, 0, 1, 14>(in, out, cyc);
  run<2, 0, 0, 2, 14>(in, out, cyc); run<4, 0, 0, 2, 14>(in, out, cyc); run<2, 1, 0, 2, 14>(in, out, cyc); run<4, 1, 0, 2, 14>(in, out, cyc);
  // half and double the VALU work
  run<2, 0, 0, 0, 7>(in, out, cyc); run<4, 0, 0, 0, 7>(in, out, cyc); run<2, 0, 0, 0, 28>(in, out, cyc); run<4, 0, 0, 0, 28>(in, out, cyc);
  run<1, 0, 0, 0, 0>(in, out, cyc); run<2, 0, 0, 0, 0>(in, out, cyc);
  return 0;
}
