set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_attention_gpu.py -x -q -k "mha" 2>&1 | tail -15
timeout 300 env ADVMIL_ATTN_BWD=two python tools/attn_bench.py 2048 16 0.25 10
timeout 300 env ADVMIL_ATTN_BWD=one python tools/attn_bench.py 2048 16 0.25 10
