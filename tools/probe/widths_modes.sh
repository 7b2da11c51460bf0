# the ESAT other-widths test under both arithmetic modes and both attention backward forms (conftest pins exact; here the mode is forced)
cd $GRAFT_REPO_ROOT
for m in exact bf16x3; do for f in two one; do
  r=$(ADVMIL_FORCE_GEMM_MODE=$m ADVMIL_ATTN_BWD=$f timeout 600 python - <<PY 2>&1 | tail -4 | tr '\n' ' ' | cut -c1-400
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["ADVMIL_GEMM_MODE"] = os.environ["ADVMIL_FORCE_GEMM_MODE"]
import torch
from advmil_amd import ops
ops.set_gemm_mode(os.environ["ADVMIL_FORCE_GEMM_MODE"])
import tests.test_parity_gpu as T
for d in (128, 256, 512):
    try:
        T.test_esat_other_backbone_widths_vs_oracle(d); print(d, "ok", end="; ")
    except AssertionError as e:
        print(d, "FAIL", str(e)[:120].replace("\n", " "), end="; ")
PY
)
  echo "$m $f: $r"
done; done
