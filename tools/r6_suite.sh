cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_suite
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -rP -p no:cacheprovider "$@" > $O/suite_full.log 2>&1
grep -E "x_storage=bf16\]" $O/suite_full.log > $O/xbf16_seen.txt
grep -vE "^\[|^$" $O/suite_full.log | tail -60 > $O/suite_tail.log
tail -5 $O/suite_tail.log
