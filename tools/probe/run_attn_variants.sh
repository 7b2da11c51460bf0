# ablation builds of the single-pass attention backward (build_alt/lib_<V>.so: attn_bwd1.hip compiled with -DB1_ABL_<V>; timing only,
# results are wrong): what the dQ stage, barrier A and the partial-tile stores cost
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "#### $v"
  ADVMIL_HIP_LIB=$GRAFT_REPO_ROOT/build_alt/lib_$v.so bash tools/probe/run_attn_prof.sh one 2>&1 | grep "attn_bwd_one\|reduce"
done
