// Dev probe: how fast can 8 waves per CU push a 256x256 fp32 C tile out in the contraction epilogue's store pattern?
//   (one store instruction = 8 rows x 128 contiguous bytes, rows N*4 bytes apart; a wave covers a 64 x 128 block in 32 stores)
//   MODE 0: stores straight from registers      MODE 1: through the per-wave LDS patch (16 ds_write_b32 + 4 ds_read_b128 per 32x32)
//   MODE 2: MODE 1 with the patch round trip but NO stores (LDS + VALU cost alone)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/store_probe tools/probe/store_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void probe(float* __restrict__ C, int64_t N, int ntn, int ntiles, float seed) {
  __shared__ float lds[8 * 32 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5, c4 = (lane & 7) * 4, rq = lane >> 3;
  float* patch = lds + (wave % 8) * 32 * 36;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int mt = t / ntn, nt = t % ntn;
    const int wr = wave / 2, wc = wave % 2;
    const int64_t rbase = (int64_t)mt * 256 + wr * 64, cbase = (int64_t)nt * 256 + wc * 128;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float res[4][4];
        if (MODE >= 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + i] = seed + r + a + b;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v4 = *reinterpret_cast<const float4*>(patch + (q * 8 + rq) * 36 + c4);
            res[q][0] = fmaxf(v4.x, 0.f); res[q][1] = fmaxf(v4.y, 0.f); res[q][2] = fmaxf(v4.z, 0.f); res[q][3] = fmaxf(v4.w, 0.f);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) { res[q][0] = seed + q; res[q][1] = seed + a; res[q][2] = seed + b; res[q][3] = seed; }
        }
        if (MODE != 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(C + (rbase + a * 32 + q * 8 + rq) * N + cbase + b * 32 + c4) = make_float4(res[q][0], res[q][1], res[q][2], res[q][3]);
        } else if (res[0][0] == 1.2345f) C[0] = res[1][1] + res[2][2] + res[3][3];
      }
  }
}

template <int MODE>
void run(const char* name, float* C, int64_t M, int64_t N, int grid) {
  const int ntn = (int)(N / 256), ntiles = (int)(M / 256) * ntn;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, 8>), dim3(grid ? grid : ntiles), dim3(512), 0, 0, C, N, ntn, ntiles, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int it = 10;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((probe<MODE, 8>), dim3(grid ? grid : ntiles), dim3(512), 0, 0, C, N, ntn, ntiles, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it, mb = (double)M * N * 4 / 1e6;
  printf("  %-52s grid %5d: %7.1f us  %6.2f TB/s\n", name, grid ? grid : ntiles, us, mb / us);
}

int main() {
  const int64_t M = 131072, N = 768;
  float* C; hipMalloc(&C, (size_t)M * N * 4 * 2);
  printf("C = %ld x %ld fp32 (%.0f MB)\n", (long)M, (long)N, (double)M * N * 4 / 1e6);
  run<0>("stores from registers, one workgroup per tile", C, M, N, 0);
  run<0>("stores from registers, persistent 256 workgroups", C, M, N, 256);
  run<1>("LDS patch + stores, one workgroup per tile", C, M, N, 0);
  run<1>("LDS patch + stores, persistent 256 workgroups", C, M, N, 256);
  run<2>("LDS patch only (no stores), persistent", C, M, N, 256);
  run<1>("LDS patch + stores, persistent 512 workgroups", C, M, N, 512);
  run<1>("LDS patch + stores, persistent 1024 workgroups", C, M, N, 1024);
  return 0;
}
