cd "${GRAFT_REPO_ROOT:-.}"
L="cluster 600 2928 2768 352 624 3920"
run() { echo "#### $*"; env GC_OFF_ONLY=1 "$@" timeout 300 python tools/probe/grad_case_check.py $L 2>&1 | grep -v amdgpu.ids | cut -c1-150; }
run GC_MODES=bf16x3 ADVMIL_SLAB_PLANES_ANY=0
run GC_MODES=bf16x3 ADVMIL_TN_PLANES=0
run GC_MODES=bf16x3 ADVMIL_DG_PLANES_ONLY=0
run GC_MODES=bf16x3 ADVMIL_ACT_BWD_IN_DH=0
run GC_MODES=bf16x3 ADVMIL_PLANES=0
run GC_MODES=exact ADVMIL_SMALL_LINEAR=0
run GC_MODES=exact ADVMIL_DEFER_SUMS=0
run GC_MODES=exact ADVMIL_GHEAD=0 ADVMIL_DTAIL=0
