# Final evidence, part B: step profiles + timelines (5 workloads), the resident product loop's kernel profile, kernel stats of the bench runs,
# pooling kernels (stats + FETCH / WRITE passes), attention core (stats + SQ pass).
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/final_round.sh steps
bash tools/r6_prof_loop.sh
ROUND=r06 bash tools/profile_round.sh bench > /dev/null
ROUND=r06 bash tools/profile_round.sh pool > /dev/null
ROUND=r06 bash tools/profile_round.sh attn > /dev/null
ls gpurun_out/prof_r06
