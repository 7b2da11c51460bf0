cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_tail_gpu.py tests/test_parity_gpu.py -q -m gpu 2>&1 | tail -2
timeout 200 python tools/probe/tail_time.py 2>&1 | grep -v amdgpu.ids | tail -5
for b in 2 1 16; do timeout 300 python bench.py --steps 200 --bags $b --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b-bag', d['ms_per_step'])"; done
