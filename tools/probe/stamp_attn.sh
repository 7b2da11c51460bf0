#!/bin/bash
# Dev probe: section timeline of the fused attention forward (shader-clock stamps of waves 0 and 4 of workgroup 0), from a build of
# attn.hip with -DAT_STAMP into tools/probe/abl/lib_attn_stamp.so. usage: stamp_attn.sh build (here) | run (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=advmil_amd/csrc
P=tools/probe/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$C -mllvm -pragma-unroll-threshold=100000"
if [ "$1" = build ]; then
  mkdir -p $P
  /opt/rocm/bin/hipcc $FLAGS -DAT_STAMP -c $C/attn.hip -o $P/attn_stamp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib_attn_stamp.so $C/gemm_f32.o $P/attn_stamp.o $C/pool.o $C/optim.o $C/graph.o $C/evalk.o
  ls -la $P/lib_attn_stamp.so
else
  ADVMIL_HIP_LIB=$PWD/$P/lib_attn_stamp.so python3 tools/probe/stamp_attn_time.py "${@:2}"
fi
