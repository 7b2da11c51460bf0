cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mid_r06
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/mid_r06/gpu_suite.log 2>&1; tail -3 gpurun_out/mid_r06/gpu_suite.log
for form in one two; do ADVMIL_ATTN_BWD=$form timeout 600 python tools/probe/attn_fuzz.py 150 611 > gpurun_out/mid_r06/attn_fuzz_$form.txt 2>&1; echo "$form: $(tail -1 gpurun_out/mid_r06/attn_fuzz_$form.txt)"; done
