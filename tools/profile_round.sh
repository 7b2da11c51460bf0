#!/bin/bash
# rocprofv3 evidence for the numbers bench.py prints (run on the GPU box through gpurun; outputs under gpurun_out/prof_<round>/ (ROUND=r06 by default)).
# Kernel-trace + stats runs and PMC runs are SEPARATE invocations (the pool refuses --pmc combined with API traces).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof_${ROUND:-r06}
mkdir -p $O
what="${1:-all}"
if [ "$what" = all ] || [ "$what" = bench ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_abmil -- python3 bench.py --steps 30 --no-extras --no-cpu-baseline > $O/bench_abmil.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_esat32k -- python3 bench.py --mode patch --patches 32768 --pool 16 --steps 10 --no-extras --no-cpu-baseline > $O/bench_esat32k.log 2>&1
fi
if [ "$what" = all ] || [ "$what" = pool ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pool16 -- python3 tools/pool_bench.py 8192 16 40 > $O/pool16.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pool1 -- python3 tools/pool_bench.py 8192 1 200 > $O/pool1.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pool16_fetch -- python3 tools/pool_bench.py 8192 16 12 > $O/pool16_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pool16_write -- python3 tools/pool_bench.py 8192 16 12 > $O/pool16_write.log 2>&1
fi
if [ "$what" = all ] || [ "$what" = gemm ]; then
  for shape in "131072 768 384" "131072 384 1024" "131072 512 1024"; do
    tag=$(echo $shape | tr ' ' x)
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gemm_${tag}_fetch -- python3 tools/pmc_gemm.py $shape 1 1 8 > $O/gemm_${tag}_fetch.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gemm_${tag}_write -- python3 tools/pmc_gemm.py $shape 1 1 8 > $O/gemm_${tag}_write.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/gemm_${tag}_sq -- python3 tools/pmc_gemm.py $shape 1 1 6 > $O/gemm_${tag}_sq.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/gemmpl_${tag}_sq -- python3 tools/pmc_gemm.py $shape 1 1 6 1 > $O/gemmpl_${tag}_sq.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gemmpl_${tag}_fetch -- python3 tools/pmc_gemm.py $shape 1 1 8 1 > $O/gemmpl_${tag}_fetch.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gemmpl_${tag}_write -- python3 tools/pmc_gemm.py $shape 1 1 8 1 > $O/gemmpl_${tag}_write.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemm_${tag}_stats -- python3 tools/pmc_gemm.py $shape 1 1 12 > $O/gemm_${tag}_stats.log 2>&1
    ADVMIL_GEMM_MODE=bf16x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemmpl_${tag}_stats -- python3 tools/pmc_gemm.py $shape 1 1 12 1 > $O/gemmpl_${tag}_stats.log 2>&1
  done
fi
if [ "$what" = all ] || [ "$what" = graph ]; then
  # PatchGCN (configs[4]'s backbone) at a size one GPU steps through: the step's kernels incl. genconv_fwd128 / genconv_bwd128
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_patchgcn -- python3 bench.py --mode graph --patches 4096 --pool 32 --steps 20 --no-extras --no-cpu-baseline > $O/bench_patchgcn.log 2>&1
fi
if [ "$what" = all ] || [ "$what" = genconv ]; then
  # GENConv aggregation kernels alone (configs[4]'s sparse gather) at the step's block-diagonal graph
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/genconv -- python3 tools/graph_bench.py 4096 16 20 > $O/genconv.log 2>&1
fi
if [ "$what" = all ] || [ "$what" = attn ]; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn2048 -- python3 tools/attn_bench.py 2048 16 0.25 10 > $O/attn2048.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/attn2048_sq -- python3 tools/attn_bench.py 2048 16 0.25 4 > $O/attn2048_sq.log 2>&1
  # the two-launch backward beside the single-pass one (the default since round 6): same tool, ADVMIL_ATTN_BWD=two in the profiler's own environment
  export ADVMIL_ATTN_BWD=two
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attn2048_two -- python3 tools/attn_bench.py 2048 16 0.25 10 > $O/attn2048_two.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/attn2048_two_sq -- python3 tools/attn_bench.py 2048 16 0.25 4 > $O/attn2048_two_sq.log 2>&1
  unset ADVMIL_ATTN_BWD
fi
# compact listing of what was produced (the CSVs themselves are merged back under gpurun_out/)
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | sort
for f in $(find $O -name "*kernel_stats.csv" | sort); do echo "== $f"; head -12 $f | cut -c1-200; done
