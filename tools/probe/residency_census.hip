// Dev probe: how many 512-thread workgroups with LDS bytes of static shared memory does a gfx950 CU hold at once?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/residency_census tools/probe/residency_census.hip && tools/probe/residency_census
// Every workgroup bumps a counter keyed by (XCC_ID, HW_ID se/sh/cu bits), spins ~30 us, and drops it again; the maximum seen per key
// is the number of co-resident workgroups. The API's answer (hipOccupancyMaxActiveBlocksPerMultiprocessor) is printed beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

template <int LDS>
__global__ __launch_bounds__(512, 4) void census(int* cnt, int* mx, float* sink, long long spin) {
  __shared__ float sm[LDS / 4];
  sm[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int key = (int)(((xcc & 15u) << 8) | ((hw >> 8) & 0xffu));
  if (threadIdx.x == 0) {
    const int now = atomicAdd(&cnt[key], 1) + 1;
    atomicMax(&mx[key], now);
  }
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < spin) acc += sm[(threadIdx.x * 7) & (LDS / 4 - 1)];
  __syncthreads();
  if (threadIdx.x == 0) atomicSub(&cnt[key], 1);
  if (acc == 123.456f) sink[0] = acc;
}

template <int LDS>
void run(int* cnt, int* mx, float* sink) {
  hipMemset(cnt, 0, 4096 * 4); hipMemset(mx, 0, 4096 * 4);
  int api = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census<LDS>, 512, 0);
  hipLaunchKernelGGL(census<LDS>, dim3(2048), dim3(512), 0, 0, cnt, mx, sink, 3000LL);     // wall_clock64: 100 MHz -> 30 us
  hipDeviceSynchronize();
  static int h[4096];
  hipMemcpy(h, mx, sizeof(h), hipMemcpyDeviceToHost);
  int hist[16] = {0}, keys = 0;
  for (int i = 0; i < 4096; ++i) if (h[i]) { ++keys; hist[h[i] < 15 ? h[i] : 15]++; }
  printf("LDS %6d B per workgroup: API says %d per CU; census over %d CU keys: ", LDS, api, keys);
  for (int i = 1; i < 16; ++i) if (hist[i]) printf("%d CUs held %d  ", hist[i], i);
  printf("\n");
}

int main() {
  int *cnt, *mx; float* sink;
  hipMalloc(&cnt, 4096 * 4); hipMalloc(&mx, 4096 * 4); hipMalloc(&sink, 64);
  run<16384>(cnt, mx, sink);
  run<32768>(cnt, mx, sink);
  run<65536>(cnt, mx, sink);
  return 0;
}
