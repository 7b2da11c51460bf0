#!/bin/bash
# Dev probe: where does the plane-fed contraction kernel's time go? Builds variants of libadvmil_hip.so from a patched COPY of
# gemm_f32.hip (the product source carries no ablation switches) and times them in one call:
#   base | noepi (epilogue replaced by a sink) | nodma (no LDS-DMA inside the K loop) | noepi+nodma (LDS reads + MFMA + barriers only)
#   | noloop (one k chunk: prologue + epilogue only)
# usage (on the GPU box): bash tools/probe/ablate_gemm.sh build   (here, cross-compile)  /  bash tools/probe/ablate_gemm.sh run
set -e
cd "$(dirname "$0")/../.."
C=advmil_amd/csrc
P=tools/probe/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$C -mllvm -pragma-unroll-threshold=100000"
if [ "$1" = build ]; then
  mkdir -p $P
  make -s -C $C
  for v in base noepi nodma noepi_nodma noloop; do
    cp $C/gemm_f32.hip $P/gemm_$v.hip
    case $v in *noepi*)
      python3 - $P/gemm_$v.hip <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old="  gemm_epilogue<TM, TN, WR, WC>(g, acc, smem, wave, lane, wr, wc, m0, n0, 0, nt_i);\n}\n\n// split-K reduction"
new="""  { float s_ = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_ += acc[a][b][r];
    if (s_ == 1.2345f) g.C[0] = s_; }
}

// split-K reduction"""
assert old in s
open(p,"w").write(s.replace(old,new))
PY
    ;; esac
    case $v in *nodma*)
      python3 - $P/gemm_$v.hip <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
for old in ("      if (k0 + 2 * BKT < K) dma(cur == 0 ? 2 : cur - 1, k0 + 2 * BKT);\n", "      if (k0 + BKT < K) dma(cur ^ 1, k0 + BKT);\n"):
    assert old in s
    s=s.replace(old, "")
open(p,"w").write(s)
PY
    ;; esac
    case $v in *noloop*)
      python3 - $P/gemm_$v.hip <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
old="  const int64_t K = g.K;\n  dma(0, 0);"
assert old in s
s=s.replace(old,"  const int64_t K = 32;\n  dma(0, 0);")
open(p,"w").write(s)
PY
    ;; esac
    /opt/rocm/bin/hipcc $FLAGS -c $P/gemm_$v.hip -o $P/gemm_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib_$v.so $P/gemm_$v.o $C/attn.o $C/pool.o $C/optim.o $C/graph.o $C/evalk.o
    rm -f $P/gemm_$v.hip
  done
  ls -la $P
else
  for v in base noepi nodma noepi_nodma noloop; do
    echo "== $v"
    ADVMIL_HIP_LIB=$PWD/$P/lib_$v.so python3 tools/probe/ablate_gemm_time.py
  done
fi
