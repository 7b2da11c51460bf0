#!/bin/bash
# PMC passes for the fused attention kernels (separate --pmc runs; no trace options beside them). usage: attn_pmc.sh OUTDIR [L] [bags] [p]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=${1:-gpurun_out/attn_pmc}; L=${2:-2048}; G=${3:-16}; P=${4:-0.25}
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 tools/attn_bench.py $L $G $P 3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/b -- python3 tools/attn_bench.py $L $G $P 3 > $O/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/c -- python3 tools/attn_bench.py $L $G $P 3 > $O/c.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    print("  " + "  ".join(f"{c}={m[c]:.4g}" for c in sorted(m)))
    if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"]:
        print(f"  VALU/MFMA={m['SQ_INSTS_VALU']/m['SQ_INSTS_MFMA']:.2f}  mfma_busy/busy_cycles={m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(m.get('SQ_BUSY_CYCLES',1),1):.3f}"
              f"  wait_any/wave={m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.3f}  wait_inst/wave={m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.3f}"
              f"  active/wave={m.get('SQ_ACTIVE_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.3f}")
    if "SQ_LDS_IDX_ACTIVE" in m:
        print(f"  lds_conflict/idx_active={m['SQ_LDS_BANK_CONFLICT']/max(m['SQ_LDS_IDX_ACTIVE'],1):.3f}")
PY
