cd "${GRAFT_REPO_ROOT:-.}"
run() { env "$@" timeout 300 python bench.py --steps 200 --bags $BAGS --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$BAGS-bag', '$*', d['ms_per_step'])"; }
for BAGS in 2 1; do
run X=0
run ADVMIL_NT_PLANES_MIN_TILES=128
run ADVMIL_NT_PLANES_MIN_TILES=64
run ADVMIL_NT_PLANES_MIN_TILES=64 ADVMIL_TWO_LAYERS_MIN_TILES=64
run ADVMIL_NT_PLANES_MIN_TILES=32 ADVMIL_TWO_LAYERS_MIN_TILES=32
run X=0
done
