// Probe of ds_read_b64_tr_b16 semantics on gfx950: every lane passes its own 8-byte-aligned LDS address; which 4 halfwords
// does it get back? LDS is filled with lds[h] = h (halfword index), so outputs identify their source directly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
#define LDS_AS __attribute__((address_space(3)))
__global__ void probe(const int* __restrict__ addr_hw, unsigned short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int h = threadIdx.x; h < 4096; h += 64) lds[h] = (unsigned short)h;
  __syncthreads();
  const int l = threadIdx.x;
  auto p = (LDS_AS bf16x4_t*)((LDS_AS unsigned short*)lds + addr_hw[l]);
  union { bf16x4_t v; unsigned short u[4]; } r;
  r.v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r.u[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: lane l -> halfword 4*l*? distinct base per lane: addr = l*64 (each lane its own 64-halfword row)
    // variant 1: [k][m] image with pitch 160 halfwords: lane i of a 16-group -> row (i>>2), cols (i&3)*4 ; group g -> +g*16 cols
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) h_addr[l] = l * 64;
      else { const int i = l & 15, gq = l >> 4; h_addr[l] = (i >> 2) * 160 + (i & 3) * 4 + gq * 16; }
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d addr %4d ->", l, h_addr[l]);
      for (int j = 0; j < 4; ++j) {
        const int h = h_out[l * 4 + j];
        if (variant == 0) printf("  (lane %2d, e%d)", h / 64, h % 64); else printf("  (k%d, m%2d)", h / 160, h % 160);
      }
      printf("\n");
    }
  }
  return 0;
}
