cd "${GRAFT_REPO_ROOT:-.}"
# resident product loop (every bag out of the device cache), ms per step, under the staging switches; same box
run() { env "$@" timeout 300 python tools/probe/resident_epoch.py 30 2>&1 | grep "resident epoch"; }
run ADVMIL_STAGE_PLANES_ONLY=0 ADVMIL_STAGE_BLOCKS=4096
run ADVMIL_STAGE_PLANES_ONLY=0
run X=0
run ADVMIL_STAGE_ABLATE=skip
run X=0
timeout 200 python bench.py --steps 100 --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph replay, 16 x 8192', d['ms_per_step'])"
