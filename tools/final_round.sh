#!/bin/bash
# Everything judged for the round, at ONE commit (.commit_hash travels with the snapshot): tools/evidence_round.sh (suite, bench lines, two-rank
# path), tools/step_profiles.sh (kernel-by-kernel profiles + timelines of the captured step), tools/profile_round.sh (rocprofv3 stats + PMC).
# usage (GPU box): tools/final_round.sh [part ...]   parts: evidence steps prof fuzz (default: evidence steps prof)
cd "${GRAFT_REPO_ROOT:-.}"
parts="${@:-evidence steps prof}"
for p in $parts; do
  case $p in
    evidence) bash tools/evidence_round.sh gpurun_out/evidence_r06 ;;
    steps) bash tools/step_profiles.sh gpurun_out/steps_r06 ;;
    prof) ROUND=r06 bash tools/profile_round.sh bench; ROUND=r06 bash tools/profile_round.sh pool; ROUND=r06 bash tools/profile_round.sh gemm ;;
    fuzz) bash tools/fuzz_round.sh gpurun_out/fuzz_r06 100 ;;
  esac
done
