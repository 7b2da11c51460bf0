cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_prof_$1
mkdir -p $O
shift
run() {
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --no-extras --no-roofline --no-cpu-baseline "$@" > $O/$tag.log 2>&1
  f=$(ls -t $O/$tag/*/*_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/step_profile.py $f 10 70 > $O/step_profile_$tag.txt && python3 tools/step_timeline.py $f > $O/timeline_$tag.txt && head -1 $O/step_profile_$tag.txt
  rm -rf $O/$tag
}
run abmil --steps 30
run bags2 --steps 60 --bags 2
