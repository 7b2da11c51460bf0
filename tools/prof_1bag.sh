cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_prof_1bag
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --no-extras --no-roofline --no-cpu-baseline --bags 1 --steps 60 > $O/t.log 2>&1
f=$(ls -t $O/t/*/*_kernel_trace.csv 2>/dev/null | head -1)
python3 tools/step_profile.py $f 10 70 > $O/step_profile_1bag.txt; python3 tools/step_timeline.py $f > $O/timeline_1bag.txt; head -1 $O/step_profile_1bag.txt
rm -rf $O/t
