#!/bin/bash
# Kernel-by-kernel profile of the captured step (no roofline / extras legs in the run): rocprofv3 --kernel-trace of bench.py, then
# tools/step_profile.py over the last 10 graph replays. Workloads: the headline step, one and two bags per step (SURVEY 8e's strong split
# at W = 16 / 8), ESAT 32k, PatchGCN 4096. usage (GPU box): tools/step_profiles.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=${1:-gpurun_out/steps_r06}
mkdir -p $O
run() {  # tag, bench args...
  tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --no-extras --no-roofline --no-cpu-baseline "$@" > $O/$tag.log 2>&1
  f=$(ls -t $O/$tag/*/*_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/step_profile.py $f 10 70 > $O/step_profile_$tag.txt && python3 tools/step_timeline.py $f > $O/timeline_$tag.txt && head -1 $O/step_profile_$tag.txt
  rm -rf $O/$tag
}
run abmil --steps 30
run bp1 --steps 60 --bags 1
run bags2 --steps 60 --bags 2
run esat32k --mode patch --patches 32768 --pool 16 --steps 12
run patchgcn --mode graph --patches 4096 --pool 32 --steps 20
