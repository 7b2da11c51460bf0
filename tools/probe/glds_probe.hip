// Probe: __builtin_amdgcn_global_load_lds with 16-byte pieces on gfx950 -- where does lane L's piece land in LDS?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/glds_probe.hip -o tools/probe/glds_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

__global__ void k(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 256];   // 2 waves x 1 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane L fetches the 16 bytes at src + (63 - L) * 4 dwords (a permuted source) into the wave's 1-KiB LDS piece
  const uint32_t* g = src + wave * 256 + (63 - lane) * 4;
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)(lds + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) dst[i] = lds[i];
}

int main() {
  std::vector<uint32_t> h(512), o(512);
  for (int i = 0; i < 512; ++i) h[i] = i;
  uint32_t *d, *e;
  hipMalloc(&d, 2048); hipMalloc(&e, 2048);
  hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, e);
  hipMemcpy(o.data(), e, 2048, hipMemcpyDeviceToHost);
  // expectation: LDS slot L of wave w (dwords 4L..4L+3) holds source dwords (63-L)*4 .. +3 of that wave's block
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int L = 0; L < 64; ++L)
      for (int q = 0; q < 4; ++q)
        if (o[w * 256 + L * 4 + q] != (uint32_t)(w * 256 + (63 - L) * 4 + q)) ++bad;
  printf("glds probe: %s (mismatches %d); lds[0..7] = %u %u %u %u %u %u %u %u\n", bad ? "UNEXPECTED LAYOUT" : "lane-linear destination, per-lane source: OK",
         bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  return bad != 0;
}
