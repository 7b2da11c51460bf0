cd "${GRAFT_REPO_ROOT:-.}"
# usage: AB="VAR=off-value" bash tools/probe/exp_legs.sh [bags...]: bench legs with the switch off and on, same box
run() { b=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --bags $b --no-extras --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b-bag', '$*', d['ms_per_step'])"; }
for b in ${@:-1 2 3 4 16}; do
run $b ${AB:-X=0}
run $b X=0
done
