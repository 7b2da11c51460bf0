cd "${GRAFT_REPO_ROOT:-.}"
run() { echo "== $*"; env "$@" timeout 300 python tools/probe/pad_fuzz.py 6 105 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200; }
run PAD_FUZZ_FROM=17
run PAD_FUZZ_FROM=16
run PAD_FUZZ_FROM=15
run PAD_FUZZ_FROM=12
