cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_full
mkdir -p $O
timeout 1500 python bench.py "$@" > $O/line.json 2> $O/line.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_full/line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
pl=d.get("product_loop") or {}
for k in ("eager_pcie_ragged","epoch1_fill_cache","epoch2_resident_capturing","resident_ragged","eager_resident_ragged"):
    e=pl.get(k); print(k, None if e is None else {q:e[q] for q in e if q in ("value","ms_per_step","steps_replayed_captured_eager","step_graphs_held","error")})
if "error" in pl: print(pl["error"])
print("pool in_step", (d.get("pool_roofline") or {}).get("in_step"))
print({k:(v.get("value"),v.get("ms_per_step")) for k,v in (d.get("sizes") or {}).items() if isinstance(v,dict)})
print(d.get("bp_every_batch_1"))
PY
tail -3 $O/line.err
