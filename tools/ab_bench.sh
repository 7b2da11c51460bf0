#!/bin/bash
# Same-box A/B of bench.py under environment toggles (box-to-box spread is +-5 %, so variants are only compared within one call).
# usage: tools/ab_bench.sh "VAR1=x VAR2=y" "VAR3=z" ...   (each argument = one variant's environment; "" = defaults)
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2; do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 60 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null)
    echo "round $round [$v] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "bags/s", d["ms_per_step"], "ms")')"
  done
done
