cd "${GRAFT_REPO_ROOT:-.}"
run() { env "$@" timeout 400 python bench.py --mode patch --patches 32768 --pool 16 --steps 20 --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('esat32k', '$*', d['ms_per_step'])"; }
run ADVMIL_ROW_PLANES=0
run ADVMIL_ROW_PLANES=1
run ADVMIL_ROW_PLANES=0
run ADVMIL_ROW_PLANES=1
