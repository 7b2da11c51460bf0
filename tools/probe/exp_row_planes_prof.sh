cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r5_prof_rp
mkdir -p $O
for v in 0 1; do
  ADVMIL_ROW_PLANES=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t$v -- python3 bench.py --no-extras --no-roofline --no-cpu-baseline --mode patch --patches 32768 --pool 16 --steps 12 > $O/t$v.log 2>&1
  f=$(ls -t $O/t$v/*/*_kernel_trace.csv 2>/dev/null | head -1)
  python3 tools/step_profile.py $f 10 70 > $O/step_profile_rp$v.txt; python3 tools/step_timeline.py $f > $O/timeline_rp$v.txt; head -1 $O/step_profile_rp$v.txt
  rm -rf $O/t$v
done
