# Final evidence, part A (one commit: .commit_hash): the GPU suite, the default bench line, the ESAT 32k line, 2 / 4 / 8 gloo ranks + watchdog.
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/final_round.sh evidence
