# usage: bash tools/r6_prof.sh <tag> [bench args...]  -> gpurun_out/r6_prof/step_profile_<tag>.txt, timeline_<tag>.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_prof
mkdir -p $O
tag=$1; shift
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --no-extras --no-roofline --no-cpu-baseline "$@" > $O/$tag.log 2>&1
f=$(ls -t $O/$tag/*/*_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/step_profile.py $f 10 70 > $O/step_profile_$tag.txt && python3 tools/step_timeline.py $f > $O/timeline_$tag.txt && head -1 $O/step_profile_$tag.txt
rm -rf $O/$tag
