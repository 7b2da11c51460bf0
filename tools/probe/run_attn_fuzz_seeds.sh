# the attention fuzzer on several more seeds (single-pass backward = the default), 150 cases each
cd $GRAFT_REPO_ROOT
export ADVMIL_GEMM_MODE=exact
for s in "$@"; do echo "seed $s: $(timeout 900 python tools/probe/attn_fuzz.py 150 $s 2>&1 | tail -1)"; done
