cd "${GRAFT_REPO_ROOT:-.}"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 300 python tools/probe/misc_fuzz.py 60 7 2>&1 | grep -v amdgpu.ids | tail -1
for b in 2 1 16; do timeout 300 python bench.py --steps 200 --bags $b --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b-bag', d['ms_per_step'])"; done
