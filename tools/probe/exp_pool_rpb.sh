cd "${GRAFT_REPO_ROOT:-.}"
# rows per workgroup of the pooling forward: in-bench pool call time (bench.py's pool_roofline leg times 40 back-to-back calls) and the step
run() { env "$@" timeout 300 python bench.py --steps 100 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pool_roofline']; print('$*', 'step', d['ms_per_step'], 'pool call us', p['avg_call_us'], 'frac', p['frac'], 'one-bag call us', p['one_bag'].get('avg_call_us'))"; }
run X=0
run ADVMIL_POOL_RPB=64
run ADVMIL_POOL_RPB=128
run ADVMIL_POOL_RPB=256
run X=0
