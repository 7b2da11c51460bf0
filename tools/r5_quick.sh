# usage: [AB_ENV='VAR=0 ...'] tools/r5_quick.sh <tag> [pytest args...]: selected GPU tests + short bench lines (16 / 2 / 1 bags per step); with AB_ENV
# every bench leg runs twice on the same box: with those variables set ("off") and without
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/r5_$tag
mkdir -p $O
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log; fi
leg() {  # name, bench args
  n=$1; shift
  if [ -n "$AB_ENV" ]; then
    env $AB_ENV timeout 600 python bench.py "$@" --no-extras --no-roofline --no-cpu-baseline > $O/bench${n}_off.json 2> $O/bench${n}_off.err
    python3 -c "import json; d=json.load(open('$O/bench${n}_off.json')); print('$n-bag OFF ($AB_ENV)', d['value'], d['ms_per_step'])"
  fi
  timeout 600 python bench.py "$@" --no-extras --no-roofline --no-cpu-baseline > $O/bench$n.json 2> $O/bench$n.err
  python3 -c "import json; d=json.load(open('$O/bench$n.json')); print('$n-bag', d['value'], d['ms_per_step'])"
}
leg 16 --steps 60
leg 2 --steps 200 --bags 2
leg 1 --steps 200 --bags 1
if [ -n "$ESAT" ]; then leg esat32k --mode patch --patches 32768 --pool 16 --steps 12; fi
