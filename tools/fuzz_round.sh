#!/bin/bash
# The randomised parity tools on NEW seeds (every case a fuzzer COUNTS instead of failing is appended to gpurun_out/fuzz_counted_cases.jsonl with
# its seeds). usage (GPU box): tools/fuzz_round.sh [outdir] [seed offset]; copy what is to be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-.}"
# the probes' bounds are written for the exact arithmetic as the starting mode (those that cover bf16x3 select it themselves); the library's own
# default is bf16x3 since round 6, so say so here, as tests/test_fuzz_gpu.py does
export ADVMIL_GEMM_MODE=exact
O=${1:-gpurun_out/fuzz_r06}
S=${2:-0}
mkdir -p $O
rm -f gpurun_out/fuzz_counted_cases.jsonl
echo "commit $(cat .commit_hash 2>/dev/null || echo unknown)" > $O/commit.txt
timeout 700 python tools/probe/oracle_fuzz.py 40 $((31 + S)) > $O/oracle_fuzz.txt 2>&1; tail -1 $O/oracle_fuzz.txt
timeout 700 python tools/probe/oracle_fuzz.py 40 $((47 + S)) >> $O/oracle_fuzz.txt 2>&1; tail -1 $O/oracle_fuzz.txt
timeout 300 python tools/probe/pad_fuzz.py 6 $((5 + S)) >> $O/oracle_fuzz.txt 2>&1; tail -1 $O/oracle_fuzz.txt
timeout 600 python tools/probe/gemm_fuzz.py 600 $((9 + S)) > $O/gemm_fuzz.txt 2>&1; tail -1 $O/gemm_fuzz.txt
timeout 600 python tools/probe/attn_fuzz.py 150 $((3 + S)) > $O/attn_fuzz.txt 2>&1; tail -1 $O/attn_fuzz.txt
timeout 600 python tools/probe/pool_fuzz.py 200 $((3 + S)) > $O/pool_fuzz.txt 2>&1; tail -1 $O/pool_fuzz.txt
timeout 600 python tools/probe/graph_fuzz.py 150 $((5 + S)) > $O/graph_fuzz.txt 2>&1; tail -1 $O/graph_fuzz.txt
timeout 600 python tools/probe/misc_fuzz.py 60 $((3 + S)) > $O/misc_fuzz.txt 2>&1; tail -1 $O/misc_fuzz.txt
timeout 900 python tools/probe/baseline_fuzz.py 45 $((5 + S)) > $O/baseline_fuzz.txt 2>&1; tail -1 $O/baseline_fuzz.txt
timeout 900 python tools/probe/dp_fuzz.py 12 $((5 + S)) > $O/dp_fuzz.txt 2>&1; tail -1 $O/dp_fuzz.txt
[ -f gpurun_out/fuzz_counted_cases.jsonl ] && cp gpurun_out/fuzz_counted_cases.jsonl $O/fuzz_counted_cases.jsonl || echo "no fuzz case was counted instead of failing" > $O/fuzz_counted_cases.jsonl
