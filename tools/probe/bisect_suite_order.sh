# which earlier test file leaves the state that makes tests/test_parity_gpu.py::test_esat_other_backbone_widths_vs_oracle fail in the full suite
cd $GRAFT_REPO_ROOT
T="tests/test_parity_gpu.py::test_esat_other_backbone_widths_vs_oracle"
for f in "$@"; do
  r=$(timeout 900 python -m pytest $f "$T" -q -m gpu -p no:randomly 2>&1 | grep -E "^FAILED|passed|failed" | tr '\n' ' ' | cut -c1-160)
  echo "$f -> $r"
done
