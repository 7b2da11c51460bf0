#!/bin/bash
# Round evidence that is not a rocprof profile: the full GPU suite, the default-flag bench line (+ the ESAT 32k line), the two-rank
# default-flag path on one GPU (the randomised parity tools: tools/fuzz_round.sh). The commit the tree was at is written to commit.txt. usage (GPU box): tools/evidence_round.sh [outdir]; copy what is to be judged into profiles/.
cd "${GRAFT_REPO_ROOT:-.}"
O=${1:-gpurun_out/evidence_r06}
mkdir -p $O
echo "commit $(cat .commit_hash 2>/dev/null || echo unknown)  $(date -u +%Y-%m-%dT%H:%MZ)" > $O/commit.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
timeout 900 python bench.py > $O/bench_line_abmil8k.json 2> $O/bench_line_abmil8k.err; tail -c 300 $O/bench_line_abmil8k.json; echo
timeout 600 python bench.py --mode patch --patches 32768 --pool 16 --steps 20 --no-extras > $O/bench_line_esat32k.json 2> $O/bench_line_esat32k.err
timeout 1800 bash tools/two_rank_one_gpu.sh > $O/two_rank_default_flags.log 2>&1; tail -c 400 $O/two_rank_default_flags.log; echo
