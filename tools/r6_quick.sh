# usage: bash tools/r6_quick.sh "<pytest -k expr or empty>" [bench args...]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_quick
mkdir -p $O
K="$1"; shift
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$K" 2>&1 | tail -15; fi
timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 60 "$@" > $O/line.json 2> $O/line.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_quick/line.json').read().strip().splitlines()[-1])
print(d['value'], 'bags/s', d['ms_per_step'], 'ms')
r=d.get('roofline') or {}
for k,v in (r.get('in_step',{}).get('others_us',{}) or {}).items(): print('  ', k, v)
PY
