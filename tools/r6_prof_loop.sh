cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6_prof
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/loop -- python3 tools/probe/resident_epoch.py 30 > $O/loop.log 2>&1
f=$(ls -t $O/loop/*/*_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/step_profile.py $f 10 40 > $O/step_profile_loop.txt && python3 tools/step_timeline.py $f > $O/timeline_loop.txt && head -1 $O/step_profile_loop.txt
tail -2 $O/loop.log
rm -rf $O/loop
