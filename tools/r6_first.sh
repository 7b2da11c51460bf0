cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_base
mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "bf16x_" -rP 2>&1 | grep -E "x_storage=bf16|passed|failed|^_____" > $O/xbf16_seen.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err; tail -c 600 $O/bench_line.json; echo
run() {
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --no-extras --no-roofline --no-cpu-baseline "$@" > $O/$tag.log 2>&1
  f=$(ls -t $O/$tag/*/*_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/step_profile.py $f 10 70 > $O/step_profile_$tag.txt && python3 tools/step_timeline.py $f > $O/timeline_$tag.txt && head -1 $O/step_profile_$tag.txt
  rm -rf $O/$tag
}
run abmil --steps 30
run esat32k --mode patch --patches 32768 --pool 16 --steps 12
