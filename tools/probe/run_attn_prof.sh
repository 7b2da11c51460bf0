# kernel-level durations of the backward forms of the attention core (tools/attn_bench.py under rocprofv3 --kernel-trace --stats)
# usage: run_attn_prof.sh [forms="two one"] [L=2048] [bags=16]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
FORMS=${1:-"two one"}; L=${2:-2048}; G=${3:-16}
for form in $FORMS; do
  export ADVMIL_ATTN_BWD=$form
  rm -rf /tmp/prof_$form
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$form -- python3 tools/attn_bench.py $L $G 0.25 10 2>&1 | grep "L=$L"
  f=$(find /tmp/prof_$form -name "*kernel_stats.csv" | head -1)
  echo "== $form"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'attn' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
done
