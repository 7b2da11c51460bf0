#!/bin/bash
# PMC passes over the GENConv kernels alone (tools/graph_bench.py): memory-side bytes, L2 hit rate, L1 request counts. One counter group per
# rocprofv3 run (no trace options next to --pmc). usage (GPU box): tools/graph_pmc.sh [outdir]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=${1:-gpurun_out/graph_pmc}
mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS" "TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 tools/graph_bench.py 4096 16 6 > $O/p$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "genconv" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"   {c:34s} n={len(v):3d} median {v[len(v)//2]:.4g}  min {v[0]:.4g} max {v[-1]:.4g}")
PY
