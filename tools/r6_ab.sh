# Same-box A/B of bench.py under environment toggles. usage: bash tools/r6_ab.sh "VAR=x" "VAR2=y" ...   ("" = defaults)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for round in 1 2 3; do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 60 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null)
    echo "round $round [$v] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "bags/s", d["ms_per_step"], "ms")')"
  done
done
