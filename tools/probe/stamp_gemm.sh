#!/bin/bash
# Dev probe: per-workgroup timeline of the plane-fed contraction kernel (100 MHz s_memrealtime stamps at entry / first chunk landed /
# K loop done / epilogue done), from a patched COPY of gemm_f32.hip. usage: stamp_gemm.sh build (here) | run (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=advmil_amd/csrc
P=tools/probe/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$C -mllvm -pragma-unroll-threshold=100000"
if [ "$1" = build ]; then
  mkdir -p $P
  cp $C/gemm_f32.hip $P/gemm_stamp.hip
  python3 - $P/gemm_stamp.hip <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
s=s.replace("#define GLB_AS __attribute__((address_space(1)))", '''#define GLB_AS __attribute__((address_space(1)))
__device__ unsigned long long g_stamps[4 * 4096];
extern "C" int advmil_debug_stamps(unsigned long long* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), (size_t)n * 8); }
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)''')
old="  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n  const int i = lane & 31, hi = lane >> 5;\n  const int wr = wave / WC, wc = wave % WC;\n  const int bid = blockIdx.x;\n  int mt_i, nt_i;\n  {   // XCD-aware tile order (see gemm_f32_kernel): the n-tiles of one A row panel"
assert old in s
s=s.replace(old, "  STAMP(0);\n"+old)
old="  dma(0, 0);\n  if (NBUF == 3 && BKT < K) dma(1, BKT);               // three buffers: two chunks in flight\n  int cur = 0;\n  for (int64_t k0 = 0; k0 < K; k0 += BKT) {"
assert old in s
s=s.replace(old, old+"\n    if (k0 == BKT) STAMP(1);")
old="    if (NBUF == 3) cur = cur == 2 ? 0 : cur + 1; else cur ^= 1;\n  }\n  gemm_epilogue<TM, TN, WR, WC>(g, acc, smem, wave, lane, wr, wc, m0, n0, 0, nt_i);\n}"
assert old in s
s=s.replace(old, "    if (NBUF == 3) cur = cur == 2 ? 0 : cur + 1; else cur ^= 1;\n  }\n  STAMP(2);\n  gemm_epilogue<TM, TN, WR, WC>(g, acc, smem, wave, lane, wr, wc, m0, n0, 0, nt_i);\n  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  STAMP(3);\n}")
open(p,"w").write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -c $P/gemm_stamp.hip -o $P/gemm_stamp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib_stamp.so $P/gemm_stamp.o $C/attn.o $C/pool.o $C/optim.o $C/graph.o $C/evalk.o
  rm -f $P/gemm_stamp.hip
  ls -la $P/lib_stamp.so
else
  ADVMIL_HIP_LIB=$PWD/$P/lib_stamp.so python3 tools/probe/stamp_gemm_time.py
fi
