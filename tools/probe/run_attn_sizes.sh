# backward forms of the attention core over bag lengths (tools/attn_bench.py, back-to-back launches): where the single-pass form pays
cd $GRAFT_REPO_ROOT
for cfg in "64 64" "128 64" "256 64" "512 64" "1024 32" "2048 16" "2048 1" "2048 2" "4096 8" "8192 4"; do
  set -- $cfg
  for form in two one; do
    echo -n "$form  "; ADVMIL_ATTN_BWD=$form timeout 300 python tools/attn_bench.py $1 $2 0.25 10 2>/dev/null | sed 's/.*bwd (prep/bwd (prep/; s/^/L='$1' bags='$2' /'
  done
done
