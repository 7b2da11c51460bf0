#!/bin/bash
# Round evidence that is not a rocprof profile: the full GPU suite, the default-flag bench line (+ the ESAT 32k line), the randomised parity
# tools (every case a fuzzer COUNTS instead of failing is appended to gpurun_out/fuzz_counted_cases.jsonl with its seeds), the two-rank
# default-flag path on one GPU. usage (GPU box): tools/evidence_round.sh [outdir]; copy what is to be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-.}"
O=${1:-gpurun_out/evidence_r04}
mkdir -p $O
rm -f gpurun_out/fuzz_counted_cases.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
timeout 900 python bench.py > $O/bench_line_abmil8k.json 2> $O/bench_line_abmil8k.err; tail -c 300 $O/bench_line_abmil8k.json; echo
timeout 600 python bench.py --mode patch --patches 32768 --pool 16 --steps 20 --no-extras > $O/bench_line_esat32k.json 2> $O/bench_line_esat32k.err
timeout 600 python tools/probe/oracle_fuzz.py 40 11 > $O/oracle_fuzz.txt 2>&1; tail -1 $O/oracle_fuzz.txt
timeout 600 python tools/probe/oracle_fuzz.py 40 23 >> $O/oracle_fuzz.txt 2>&1; tail -1 $O/oracle_fuzz.txt
timeout 300 python tools/probe/pad_fuzz.py 6 3 >> $O/oracle_fuzz.txt 2>&1; tail -1 $O/oracle_fuzz.txt
timeout 600 python tools/probe/gemm_fuzz.py 600 7 > $O/gemm_fuzz.txt 2>&1; tail -1 $O/gemm_fuzz.txt
timeout 600 python tools/probe/attn_fuzz.py 150 2 > $O/attn_fuzz.txt 2>&1; tail -1 $O/attn_fuzz.txt
timeout 600 python tools/probe/pool_fuzz.py 200 2 > $O/pool_fuzz.txt 2>&1; tail -1 $O/pool_fuzz.txt
timeout 600 python tools/probe/graph_fuzz.py 150 4 > $O/graph_fuzz.txt 2>&1; tail -1 $O/graph_fuzz.txt
timeout 600 python tools/probe/misc_fuzz.py 60 2 > $O/misc_fuzz.txt 2>&1; tail -1 $O/misc_fuzz.txt
timeout 900 python tools/probe/baseline_fuzz.py 45 3 > $O/baseline_fuzz.txt 2>&1; tail -1 $O/baseline_fuzz.txt
timeout 900 python tools/probe/dp_fuzz.py 12 2 > $O/dp_fuzz.txt 2>&1; tail -1 $O/dp_fuzz.txt
[ -f gpurun_out/fuzz_counted_cases.jsonl ] && cp gpurun_out/fuzz_counted_cases.jsonl $O/fuzz_counted_cases.jsonl || echo "no fuzz case was counted instead of failing" > $O/fuzz_counted_cases.jsonl
timeout 600 bash tools/two_rank_one_gpu.sh > $O/two_rank_default_flags.log 2>&1; tail -c 400 $O/two_rank_default_flags.log; echo
