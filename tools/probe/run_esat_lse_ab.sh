# same-box A/B of the ESAT 32k step with / without the log-sum-exp memo of the attention forward (advmil_mha_fwd_lse)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in 0 1; do
  echo -n "ADVMIL_MHA_LSE_MEMO=$v  "
  ADVMIL_MHA_LSE_MEMO=$v timeout 600 python bench.py --mode patch --patches 32768 --pool 16 --steps 60 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{}); print('ms_per_step', d['ms_per_step'], ' attention in step: fwd', r.get('fwd_launch_us'), 'bwd', r.get('bwd_launches_us'), 'frac', r.get('frac'))
"
done; done
