# same-box A/B of the ESAT 32k step (BASELINE configs[3]) under the two backward forms of the attention core
# usage: run_esat_ab.sh [steps=60] [rounds=3]
cd $GRAFT_REPO_ROOT
STEPS=${1:-60}; ROUNDS=${2:-3}
for r in $(seq $ROUNDS); do for form in two one; do
  echo -n "ADVMIL_ATTN_BWD=$form  "
  ADVMIL_ATTN_BWD=$form timeout 600 python bench.py --mode patch --patches 32768 --pool 16 --steps $STEPS --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('ms_per_step', d['ms_per_step'], ' attention in step: fwd', r.get('fwd_launch_us'), 'us  bwd', r.get('bwd_launches_us'), 'us  frac', r.get('frac'), ' back to back', r.get('back_to_back',{}).get('frac'))
"
done; done
