// Dev probe: what does the matrix pipe deliver for the bf16x3 inner loop's instruction mix on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_probe tools/probe/mfma_probe.hip && tools/probe/mfma_probe
// Variants (per 24-MFMA phase = 8 accumulators x 3 products, the 256x256 tile's per-wave 16-deep k step):
//   MODE 0: MFMAs only, operands fixed in registers
//   MODE 1: + 12 ds_read_b128 per phase feeding the operands (conflict-free addresses), no barrier
//   MODE 2: + one s_barrier per phase (all waves in lockstep)
//   MODE 3: two wave groups in anti-phase (reads | barrier | MFMAs | barrier), group 1 one barrier behind
//   MODE 4: MODE 2 + LDS-DMA of the next chunk (8 x global_load_lds_dwordx4 per wave per 2 phases = 64 KB per workgroup), all 8
//           pieces issued in one burst behind the barrier of every second phase (the product kernel's placement)
//   MODE 5: same bytes, 2 pieces behind every 12 MFMAs
//   MODE 6: MODE 3 (anti-phase) + 4 pieces per phase issued in the read interval
//   MODE 7: MODE 3 + 4 pieces per phase issued behind barrier A (ahead of the MFMAs)
// DATA: 0 = zero operands, 1 = random-ish operands (switching power lowers the clock)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
union Frag { uint4 u; bf16x8 v; };

template <int MODE, int WAVES, int SRC = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void probe(const uint4* __restrict__ src, float* __restrict__ out, int phases, const uint4* __restrict__ big) {
  __shared__ __attribute__((aligned(16))) uint4 lds[8192];   // 128 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 64 * WAVES) lds[i] = src[i];
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  Frag f[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) f[j].u = lds[(wave * 12 + j) * 64 + lane];
  const int grp = wave >> 2;
  constexpr bool STAGM = (MODE == 3 || MODE == 6 || MODE == 7);
  // DMA: the workgroup's private 64 KB region of `big` (L2 resident), 1 KB per piece, landing in the upper 64 KB of lds
  // SRC 0: a piece = 1 KB contiguous; SRC 1: 16 rows x 64 B out of 2 KB rows (the product kernel's pieces: half cache lines);
  // SRC 2: 8 rows x 128 B out of 2 KB rows (full cache lines). The workgroup's region stays L2-sized in every form.
  const uint4* gsrc = SRC == 0 ? big + (size_t)blockIdx.x * 4096 + wave * 512 + lane
                    : SRC == 1 ? big + (size_t)blockIdx.x * 4096 * 4 + (size_t)(wave * 16 + (lane >> 2)) * 128 + (lane & 3)
                               : big + (size_t)blockIdx.x * 4096 * 4 + (size_t)(wave * 8 + (lane >> 3)) * 128 + (lane & 7);
  auto piece = [&](int j) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (SRC == 0 ? j * 64 : (SRC == 1 ? j * 4 : j * 8))),
                                     (__attribute__((address_space(3))) void*)(lds + 4096 + wave * 512 + j * 64), 16, 0, 0);
  };
  if (STAGM && grp == 1) __builtin_amdgcn_s_barrier();
  for (int p = 0; p < phases; ++p) {
    if (MODE >= 1) {
      const int base = ((p & 3) * 96 + (wave & 7) * 12) * 64 + lane;    // 16 B per lane, lane-linear: conflict-free
#pragma unroll
      for (int j = 0; j < 12; ++j) f[j].u = lds[(base + j * 64) & 8191];
    }
    if (MODE == 6) {
#pragma unroll
      for (int j = 0; j < 4; ++j) piece((p & 1) * 4 + j);
    }
    if (MODE >= 4 && (p & 1) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    if (MODE == 4 && (p & 1) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) piece(j);
    }
    if (MODE == 7) {
#pragma unroll
      for (int j = 0; j < 4; ++j) piece((p & 1) * 4 + j);
    }
    if (STAGM) __builtin_amdgcn_s_setprio(1);
    // A frags: f[0..3] (2 row tiles x hi/lo), B frags: f[4..11] (4 col tiles x hi/lo)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (MODE == 5) { piece((p & 1) * 4 + a * 2); piece((p & 1) * 4 + a * 2 + 1); }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a * 2 + 1].v, f[4 + b * 2].v, acc[a * 4 + b], 0, 0, 0);
        acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a * 2].v, f[4 + b * 2 + 1].v, acc[a * 4 + b], 0, 0, 0);
        acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a * 2].v, f[4 + b * 2].v, acc[a * 4 + b], 0, 0, 0);
      }
    }
    if (STAGM) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  }
  if (STAGM && grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 1.2345f) out[tid] = s;
}

template <int MODE, int WAVES, int SRC = 0>
void run(const char* name, const uint4* src, float* out, int blocks, const uint4* big) {
  const int phases = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, WAVES, SRC>), dim3(blocks), dim3(64 * WAVES), 0, 0, src, out, 64, big);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, WAVES, SRC>), dim3(blocks), dim3(64 * WAVES), 0, 0, src, out, phases, big);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * WAVES * phases * 24;
  const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
  printf("  %-44s blocks %4d waves/WG %d: %8.3f ms  %7.0f TF raw bf16 = %.2f of 2500\n", name, blocks, WAVES, ms, tf, tf / 2500.0);
}

int main() {
  uint4* src; float* out;
  hipMalloc(&src, 8192 * 16); hipMalloc(&out, 4096);
  uint4* big; hipMalloc(&big, (size_t)256 * 65536 * 4); hipMemset(big, 0, (size_t)256 * 65536 * 4);
  for (int data = 0; data < 2; ++data) {
    std::vector<uint32_t> h(8192 * 4);
    uint32_t x = 12345;
    for (auto& v : h) {
      x = x * 1664525u + 1013904223u;
      // two bf16 in [-1, 1): sign + exponent 0x3f0..0x3f7 region
      const uint32_t lo = 0x3c00u + ((x >> 8) & 0x3ffu) + ((x >> 3) & 0x8000u), hi = 0x3c00u + ((x >> 20) & 0x3ffu) + ((x >> 1) & 0x8000u);
      v = data ? (lo | (hi << 16)) : 0u;
    }
    hipMemcpy(src, h.data(), 8192 * 16, hipMemcpyHostToDevice);
    printf("== operands: %s\n", data ? "random" : "zero");
    run<0, 4>("MFMA only, 1 wave/SIMD", src, out, 256, big);
    run<0, 8>("MFMA only, 2 waves/SIMD", src, out, 256, big);
    run<1, 8>("MFMA + 12 ds_read_b128 / 24 MFMA", src, out, 256, big);
    run<2, 8>("... + s_barrier per phase (lockstep)", src, out, 256, big);
    run<3, 8>("two groups in anti-phase (2 barriers / phase)", src, out, 256, big);
    run<4, 8>("lockstep + DMA burst behind barrier", src, out, 256, big);
    run<5, 8>("lockstep + DMA 2 pieces / 12 MFMA", src, out, 256, big);
    run<4, 8, 1>("lockstep + DMA burst, 16 rows x 64 B pieces", src, out, 256, big);
    run<4, 8, 2>("lockstep + DMA burst, 8 rows x 128 B pieces", src, out, 256, big);
    run<5, 8, 1>("lockstep + DMA spread, 16 rows x 64 B pieces", src, out, 256, big);
    run<5, 8, 2>("lockstep + DMA spread, 8 rows x 128 B pieces", src, out, 256, big);
    run<6, 8>("anti-phase + 4 pieces in read interval", src, out, 256, big);
    run<7, 8>("anti-phase + 4 pieces behind barrier A", src, out, 256, big);
    run<1, 4>("MFMA + reads, 1 wave/SIMD", src, out, 256, big);
    run<2, 4>("MFMA + reads + barrier, 1 wave/SIMD", src, out, 256, big);
  }
  return 0;
}
