cd "${GRAFT_REPO_ROOT:-.}"
run() { echo "== $*"; env "$@" timeout 300 python tools/probe/pad_fuzz.py 6 105 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200; }
run X=0
run ADVMIL_SLAB_PLANES_ANY=0
run ADVMIL_TN_PLANES_MINK=1024
run ADVMIL_GHEAD=0
run ADVMIL_TN_GROUP=0
run ADVMIL_DX_CHAIN=0
run ADVMIL_DTAIL=0
run ADVMIL_DEFER_SUMS=0
