#!/usr/bin/env python3
"""Copy the rocprofv3 summaries produced by tools/profile_round.sh (gpurun_out/prof_r02/...) into profiles/ (tracked) and
assemble the PMC JSONs bench.py / DESIGN.md cite. FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE
counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> doubled here, stated per entry."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import prof_summary  # noqa: E402
import csv  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
DST = os.path.join(ROOT, "profiles")


def newest(pattern):
    """gpurun merges every call's outputs into the same tree, so a run directory can hold CSVs of several calls (one per profiled
    pid): only the newest one describes the current kernels."""
    f = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


first = newest


def counter(run, name):
    agg = {}
    f = newest(os.path.join(run, "**", "*counter_collection.csv"))
    if f:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != name:
                continue
            a = agg.setdefault(row["Kernel_Name"].split("(")[0].replace("void ", ""), {})
            a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in agg.items()}


def stats_avg(run):
    f = first(f"{run}/**/*kernel_stats.csv")
    out = {}
    if f:
        for row in csv.DictReader(open(f)):
            out[row["Name"].split("(")[0].replace("void ", "")] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
    return out


for run in ("bench_abmil", "bench_esat32k", "bench_patchgcn", "pool16", "pool1", "attn2048", "genconv"):
    f = first(f"{run}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"{TAG}_{run}_kernel_stats.csv"))
    lg = os.path.join(SRC, run + ".log")
    if os.path.exists(lg) and run.startswith("bench"):
        shutil.copy(lg, os.path.join(DST, f"{TAG}_{run}.log"))

# ---- attention-pool kernels: HBM bytes per launch vs algorithmic bytes, per kernel, both slab sizes
fe, wr = counter("pool16_fetch", "FETCH_SIZE"), counter("pool16_write", "WRITE_SIZE")
if fe:
    rows, D = 131072, 384
    st16, st1 = stats_avg("pool16"), stats_avg("pool1")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE -- python3 tools/pool_bench.py 8192 16 12 ; separate --pmc WRITE_SIZE pass; "
                      "durations: rocprofv3 --kernel-trace --stats -- python3 tools/pool_bench.py 8192 16 40 (and 8192 1 200); h as operand planes "
                      "(pool_bench.py's default since round 6: the <.., true> instantiations)",
           "units": "FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE x2 (gfx950: wide coalesced reads tallied at half their bytes)",
           "rows": rows, "D": D, "kernels": {}}
    wanted = ("pool_partial8_online_kernel", "pool_merge_online_kernel", "pool_partial8_kernel", "pool_bwd_dot_kernel", "softmax_stats_kernel",
              "pool_bwd_ds_kernel", "colsum_merge_kernel")
    for k in sorted(fe):                      # (templated kernels appear with their arguments: <MEAN, PL> -- PL = h read as operand planes)
        base = k.split("<")[0]
        if base not in wanted:
            continue
        fetched = 2.0 * fe[k][1] * 1024
        written = wr.get(k, (0, 0.0))[1] * 1024
        alg = {"pool_partial8_kernel": 4.0 * rows * D + 4.0 * rows, "pool_partial8_online_kernel": 4.0 * rows * D + 4.0 * rows,
               "pool_bwd_dot_kernel": 4.0 * rows * D + 8.0 * rows}.get(base)
        ent = {"fetch_bytes_per_launch": fetched, "write_bytes_per_launch": written, "hbm_bytes_per_launch": fetched + written,
               "algorithmic_bytes_per_launch": alg}
        if k in st16:
            ent["avg_us_16_bags"] = st16[k][1]
            if alg:
                ent["GBps_16_bags"] = alg / st16[k][1] / 1e3
                ent["frac_of_8TBps_16_bags"] = alg / st16[k][1] / 1e3 / 8000.0
        if k in st1:
            ent["avg_us_1_bag"] = st1[k][1]
            if alg:
                ent["frac_of_8TBps_1_bag"] = (alg / 16.0) / st1[k][1] / 1e3 / 8000.0
        out["kernels"][k] = ent
    json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_pool.json"), "w"), indent=1)

# ---- attention core: instruction mix
va, mf, mb = counter("attn2048_sq", "SQ_INSTS_VALU"), counter("attn2048_sq", "SQ_INSTS_MFMA"), counter("attn2048_sq", "SQ_VALU_MFMA_BUSY_CYCLES")
if va:
    st = stats_avg("attn2048")
    out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python3 tools/attn_bench.py 2048 16 0.25 4",
           "shape": "16 bags x 2048 tokens x 8 heads x 48, attention dropout 0.25", "kernels": {}}
    for k in va:
        if not k.startswith("attn_"):
            continue
        ent = {"valu_insts": va[k][1], "mfma_insts": mf.get(k, (0, 0))[1], "mfma_busy_cycles": mb.get(k, (0, 0))[1]}
        if ent["mfma_insts"]:
            ent["valu_per_mfma"] = ent["valu_insts"] / ent["mfma_insts"]
        if k in st:
            ent["avg_us"] = st[k][1]
            ent["mfma_pipe_busy_frac_at_2.4GHz"] = ent["mfma_busy_cycles"] / (st[k][1] * 1e-6 * 2.4e9 * 1024)
        out["kernels"][k] = ent
    # the two-launch backward (ADVMIL_ATTN_BWD=two), same tool: its dQ / dK,dV kernels beside the single-pass kernel above
    va2, mf2, mb2 = counter("attn2048_two_sq", "SQ_INSTS_VALU"), counter("attn2048_two_sq", "SQ_INSTS_MFMA"), counter("attn2048_two_sq", "SQ_VALU_MFMA_BUSY_CYCLES")
    if va2:
        st2 = stats_avg("attn2048_two")
        out["two_launch_backward"] = {}
        for k in va2:
            if not (k.startswith("attn_bwd_dq") or k.startswith("attn_bwd_dkv")):
                continue
            ent = {"valu_insts": va2[k][1], "mfma_insts": mf2.get(k, (0, 0))[1], "mfma_busy_cycles": mb2.get(k, (0, 0))[1]}
            if ent["mfma_insts"]:
                ent["valu_per_mfma"] = ent["valu_insts"] / ent["mfma_insts"]
            if k in st2:
                ent["avg_us"] = st2[k][1]
            out["two_launch_backward"][k] = ent
    json.dump(out, open(os.path.join(DST, f"{TAG}_pmc_attn.json"), "w"), indent=1)

# ---- slab contractions: HBM traffic per launch + instruction mix, generic (on-the-fly split) and plane-fed kernels
def kname(sym):
    if sym.startswith("gemm_nt_planes_kernel") or sym.startswith("gemm_tn_planes_kernel"):
        return sym[:sym.index("<")] + "<%s>" % sym[sym.index("<") + 1:sym.index(">")]
    t = sym[sym.index("<") + 1:sym.index(">")].split(", ")       # A_KC, B_KC, TM, TN, SPLIT, PRE, BKT, WR, WC
    return "gemm_f32_kernel<%d,%d,%s,%s>" % (t[0] == "true", t[1] == "true", int(t[2]) * (int(t[7]) // 2), int(t[3]) * (int(t[8]) // 2))


launches, planes_launches = {}, {}
for shape in ("131072x768x384", "131072x384x1024", "131072x512x1024"):
    M, N, K = (int(v) for v in shape.split("x"))
    for prefix, dest in (("gemm", launches), ("gemmpl", planes_launches)):
        fe = counter(f"{prefix}_{shape}_fetch", "FETCH_SIZE")
        wr = counter(f"{prefix}_{shape}_write", "WRITE_SIZE")
        va, mf = counter(f"{prefix}_{shape}_sq", "SQ_INSTS_VALU"), counter(f"{prefix}_{shape}_sq", "SQ_INSTS_MFMA")
        mb, ld = counter(f"{prefix}_{shape}_sq", "SQ_VALU_MFMA_BUSY_CYCLES"), counter(f"{prefix}_{shape}_sq", "SQ_INSTS_LDS")
        st = stats_avg(f"{prefix}_{shape}_stats")
        src = fe or va or {}
        ks = [k for k in src if k.startswith("gemm_f32_kernel") or k.startswith("gemm_nt_planes_kernel") or k.startswith("gemm_tn_planes_kernel")]
        if not ks:
            continue
        k = max(ks, key=lambda n: src[n][0])
        ent = {"kernel": kname(k), "kernel_symbol": k, "gemm_mode": "bf16x3", "algorithmic_bytes_per_launch": 4.0 * (M * K + N * K + M * N),
               "flops_per_launch": 2.0 * M * N * K}
        if k in fe:
            ent["fetch_bytes_per_launch"] = 2.0 * fe[k][1] * 1024
            ent["dispatches"] = fe[k][0]
        if k in wr:
            ent["write_bytes_per_launch"] = wr[k][1] * 1024
        if "fetch_bytes_per_launch" in ent and "write_bytes_per_launch" in ent:
            ent["hbm_bytes_per_launch"] = ent["fetch_bytes_per_launch"] + ent["write_bytes_per_launch"]
        if k in va:
            ent.update(valu_insts=va[k][1], mfma_insts=mf.get(k, (0, 0))[1], lds_insts=ld.get(k, (0, 0))[1],
                       mfma_busy_cycles=mb.get(k, (0, 0))[1])
            if ent["mfma_insts"]:
                ent["valu_per_mfma"] = ent["valu_insts"] / ent["mfma_insts"]
        if k in st:
            ent["avg_us"] = st[k][1]
            ent["tflops"] = ent["flops_per_launch"] / st[k][1] / 1e6
            ent["frac_of_bf16x3_roof"] = ent["tflops"] / (2500.0 / 3.0)
            if "mfma_busy_cycles" in ent:
                ent["mfma_pipe_busy_frac_at_2.4GHz"] = ent["mfma_busy_cycles"] / (st[k][1] * 1e-6 * 2.4e9 * 1024)
            # operand bytes staged into LDS per CU clock (A and B tiles of every workgroup; what the per-CU load path carries)
        dest[shape] = ent
if launches or planes_launches:
    json.dump({"command": "ADVMIL_GEMM_MODE=bf16x3 rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_gemm.py M N K 1 1 8 [1 = operands as planes]; "
                          "separate --pmc WRITE_SIZE, --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS and "
                          "--kernel-trace --stats passes (tools/profile_round.sh gemm)",
               "units": "KB counters; FETCH_SIZE x2 (gfx950 correction)", "launches": launches, "planes_kernel_launches": planes_launches},
              open(os.path.join(DST, f"{TAG}_pmc_gemm_all.json"), "w"), indent=1)
print(sorted(os.listdir(DST)))
