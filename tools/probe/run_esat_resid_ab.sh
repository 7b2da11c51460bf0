# same-box A/B of the ESAT 32k step with / without the residual-gradient hand-over (ops.ResidualGrads)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in 0 1; do
  echo -n "ADVMIL_RESIDUAL_HANDOVER=$v  "
  ADVMIL_RESIDUAL_HANDOVER=$v timeout 600 python bench.py --mode patch --patches 32768 --pool 16 --steps 60 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', d['ms_per_step'], 'launches', d.get('launches_per_step', d.get('config',{}).get('launches_per_step')))
"
done; done
