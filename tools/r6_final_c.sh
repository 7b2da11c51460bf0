# Final evidence, part C: HBM traffic + instruction mix of the slab contractions (separate --pmc passes per counter group), PatchGCN stats.
cd "${GRAFT_REPO_ROOT:-.}"
ROUND=r06 bash tools/profile_round.sh gemm > /dev/null
ROUND=r06 bash tools/profile_round.sh graph > /dev/null
ls gpurun_out/prof_r06 | head -40
