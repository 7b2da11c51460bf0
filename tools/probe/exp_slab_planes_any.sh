cd "${GRAFT_REPO_ROOT:-.}"
run() { b=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --bags $b --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b-bag', '$*', d['ms_per_step'])"; }
for b in 1 2 4; do
run $b ADVMIL_SLAB_PLANES_ANY=0
run $b ADVMIL_SLAB_PLANES_ANY=1
run $b ADVMIL_SLAB_PLANES_ANY=1 ADVMIL_NT_PLANES_MIN_TILES=128
run $b ADVMIL_SLAB_PLANES_ANY=0
done
