#!/bin/bash
# FETCH_SIZE of the plane-fed two-layer shape (131072 x 512 x 1024) under different placements of the A operand's hi / lo planes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp ADVMIL_GEMM_MODE=bf16x3
O=${1:-gpurun_out/fetch_skew}
mkdir -p $O
for sk in -1 1 65536 1048576; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/s$sk -- python3 tools/pmc_gemm.py 131072 512 1024 1 1 16 1 $sk > $O/s$sk.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/pmc_gemm.py 131072 512 1024 1 1 16 1 -1 > $O/t.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t4096 -- python3 tools/pmc_gemm.py 131072 512 1024 1 1 16 1 1 > $O/t4096.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for sk in ("-1", "1", "65536", "1048576"):
    vals = []
    for f in glob.glob(f"{O}/s{sk}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_nt_planes" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                vals.append(2 * float(r["Counter_Value"]) * 1024 / 1e6)       # KB -> MB, x2 (gfx950 correction)
    print(f"skew {sk:>6}: FETCH per launch (MB, x2 corrected): " + " ".join(f"{v:.0f}" for v in vals) + f"   mean {sum(vals)/max(len(vals),1):.0f}  (A planes 537 MB + B 2 MB algorithmic)")
for t in ("t", "t4096"):
    for f in glob.glob(f"{O}/{t}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_nt_planes" in r["Name"]:
                print(t, r["Name"][:60], "avg us", float(r["AverageNs"]) / 1e3)
PY
