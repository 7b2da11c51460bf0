// Dev probe: do the matrix pipe and the vector ALU of a gfx950 SIMD overlap, and for which instruction kinds?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/overlap_probe tools/probe/overlap_probe.hip && tools/probe/overlap_probe
// One workgroup per CU of W waves per SIMD; every wave runs `iters` rounds of { NM independent v_mfma_f32_32x32x16_bf16 ; NV VALU
// instructions of kind KIND } written as ONE asm block per round (MFMAs first, or interleaved 1 : NV/NM), operands in registers.
// Reported: cycles per round per SIMD (s_memtime of wave 0 / rounds / W) for MFMA only, VALU only, both.
//   KIND 0 v_fma_f32   1 v_exp_f32   2 v_cndmask_b32   3 v_cvt_pk_bf16_f32   4 v_pk_add_f32   5 v_mul_lo_u32   6 v_cmp_ge_u32 sdwa
//   SPLIT 1: even waves issue only the MFMAs, odd waves only the VALU (different waves on the same SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define V_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
template <int KIND>
__device__ __forceinline__ void valu8(float (&x)[8], float a, float b) {
  if (KIND == 0)
    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                 "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
  else if (KIND == 1)
    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                 "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
  else if (KIND == 2)
    asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                 "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc");
  else if (KIND == 3)
    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\n\tv_cvt_pk_bf16_f32 %1, %1, %8\n\tv_cvt_pk_bf16_f32 %2, %2, %8\n\tv_cvt_pk_bf16_f32 %3, %3, %8\n\t"
                 "v_cvt_pk_bf16_f32 %4, %4, %8\n\tv_cvt_pk_bf16_f32 %5, %5, %8\n\tv_cvt_pk_bf16_f32 %6, %6, %8\n\tv_cvt_pk_bf16_f32 %7, %7, %8"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
  else if (KIND == 4)
    asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %2, %2, %3\n\tv_pk_add_f32 %4, %4, %5\n\tv_pk_add_f32 %6, %6, %7\n\t"
                 "v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %2, %2, %3\n\tv_pk_add_f32 %4, %4, %5\n\tv_pk_add_f32 %6, %6, %7"
                 : "+v"(*(double*)&x[0]), "+v"(*(double*)&x[2]), "+v"(*(double*)&x[4]), "+v"(*(double*)&x[6]),
                   "+v"(*(double*)&x[0]), "+v"(*(double*)&x[2]), "+v"(*(double*)&x[4]), "+v"(*(double*)&x[6]) : "v"(a), "v"(b));
  else if (KIND == 5)
    asm volatile("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\t"
                 "v_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
  else
    asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cmp_ge_u32_sdwa vcc, %1, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                 "v_cmp_ge_u32_sdwa vcc, %2, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cmp_ge_u32_sdwa vcc, %3, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                 "v_cmp_ge_u32_sdwa vcc, %4, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cmp_ge_u32_sdwa vcc, %5, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                 "v_cmp_ge_u32_sdwa vcc, %6, %8 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cmp_ge_u32_sdwa vcc, %7, %8 src0_sel:BYTE_1 src1_sel:DWORD"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc");
}

// one round: 4 MFMAs on 4 independent accumulators and 4 x 8 VALU, interleaved MFMA, 8 VALU, MFMA, ...
template <int KIND, bool DO_M, bool DO_V, int W, int SPLIT>
__global__ __launch_bounds__(256 * W) void probe(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = in[(a * 16 + r + lane) & 1023];
  union { uint4 u; bf16x8 v; } fa, fb;
  fa.u = make_uint4(__float_as_uint(in[lane]), __float_as_uint(in[lane + 64]), __float_as_uint(in[lane + 128]), __float_as_uint(in[lane + 192]));
  fb.u = make_uint4(__float_as_uint(in[lane + 256]), __float_as_uint(in[lane + 320]), __float_as_uint(in[lane + 384]), __float_as_uint(in[lane + 448]));
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[(lane * 8 + i) & 1023];
  const float ca = 0.999f, cb = 0.001f;
  const bool m_on = DO_M && (!SPLIT || (wave & 4) == 0), v_on = DO_V && (!SPLIT || (wave & 4) != 0);   // waves w and w + 4 share a SIMD
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (m_on) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[a], 0, 0, 0);
      if (v_on) valu8<KIND>(x, ca, cb);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
static double wall_ns(F launch, int iters) {        // kernel wall time per round (hipEvents), ns
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(50);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  launch(iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e6 * ms / iters;
}

template <int KIND, int W, int SPLIT>
void run(const float* in, float* out, unsigned long long* cyc, const char* name) {
  const int iters = 4000;
  // WALL time of the whole kernel per round: every SIMD carries W waves, each round = 4 MFMAs and / or 32 VALU per issuing wave.
  // (The cycle count a single wave reads back is NOT the SIMD's: the oldest wave is served first and sees little of its partners.)
  const double m = wall_ns([&](int n) { hipLaunchKernelGGL((probe<KIND, true, false, W, SPLIT>), dim3(256), dim3(256 * W), 0, 0, in, out, n, cyc); }, iters);
  const double v = wall_ns([&](int n) { hipLaunchKernelGGL((probe<KIND, false, true, W, SPLIT>), dim3(256), dim3(256 * W), 0, 0, in, out, n, cyc); }, iters);
  const double b = wall_ns([&](int n) { hipLaunchKernelGGL((probe<KIND, true, true, W, SPLIT>), dim3(256), dim3(256 * W), 0, 0, in, out, n, cyc); }, iters);
  const int wm = SPLIT ? W / 2 : W, wv = SPLIT ? W - W / 2 : W;          // waves per SIMD that issue MFMAs / VALU
  printf("%-20s W=%d split=%d : wall per round  MFMA only %6.1f ns (%5.1f per MFMA)  VALU only %6.1f ns (%4.2f per VALU)  both %6.1f ns  = %.2f x (MFMA + VALU), %.2f x max\n",
         name, W, SPLIT, m, m / (4.0 * wm), v, v / (32.0 * wv), b, b / (m + v), b / (m > v ? m : v));
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64);
  float hin[1024];
  for (int i = 0; i < 1024; ++i) hin[i] = 0.5f + 0.0001f * (float)((i * 2654435761u) % 1000);
  hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
  run<0, 1, 0>(in, out, cyc, "v_fma_f32");        run<0, 2, 0>(in, out, cyc, "v_fma_f32");        run<0, 2, 1>(in, out, cyc, "v_fma_f32");       run<0, 4, 0>(in, out, cyc, "v_fma_f32");
  run<0, 4, 1>(in, out, cyc, "v_fma_f32");
  run<1, 1, 0>(in, out, cyc, "v_exp_f32");        run<1, 2, 0>(in, out, cyc, "v_exp_f32");        run<1, 2, 1>(in, out, cyc, "v_exp_f32");
  run<2, 1, 0>(in, out, cyc, "v_cndmask_b32");    run<2, 2, 0>(in, out, cyc, "v_cndmask_b32");
  run<3, 1, 0>(in, out, cyc, "v_cvt_pk_bf16_f32"); run<3, 2, 0>(in, out, cyc, "v_cvt_pk_bf16_f32");
  run<4, 1, 0>(in, out, cyc, "v_pk_add_f32");     run<4, 2, 0>(in, out, cyc, "v_pk_add_f32");
  run<5, 1, 0>(in, out, cyc, "v_mul_lo_u32");     run<5, 2, 0>(in, out, cyc, "v_mul_lo_u32");
  run<6, 1, 0>(in, out, cyc, "v_cmp_ge_u32_sdwa"); run<6, 2, 0>(in, out, cyc, "v_cmp_ge_u32_sdwa");
  return 0;
}
