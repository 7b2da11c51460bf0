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
#define STAMP(slot, k) do { if (threadIdx.x == 0 && (slot) < 4096) g_stamps[(slot) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)''')
old="    const int i = lane & 31, hi = lane >> 5;\n    const bool has_next = v + G < ntile;"
assert old in s
s=s.replace(old, "    STAMP(v, 0);\n"+old)
old="      cur = cur == NBUF - 1 ? 0 : cur + 1;\n    }\n"
assert old in s
s=s.replace(old, "      if (c == 0) STAMP(v, 1);\n"+old+"    STAMP(v, 2);\n")
old="    mt_i = mt_n; nt_i = nt_n;\n    if (has_next) {"
assert old in s
s=s.replace(old, "    STAMP(v, 3);\n"+old)
open(p,"w").write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -c $P/gemm_stamp.hip -o $P/gemm_stamp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib_stamp.so $P/gemm_stamp.o $C/attn.o $C/pool.o $C/optim.o $C/graph.o $C/evalk.o
  rm -f $P/gemm_stamp.hip
  ls -la $P/lib_stamp.so
else
  ADVMIL_HIP_LIB=$PWD/$P/lib_stamp.so python3 tools/probe/stamp_gemm_time.py
fi
