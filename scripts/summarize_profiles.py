"""Condense gpurun_out/prof_* (rocprofv3 csv) into profiles/<tag>_*.{csv,json} summaries that get committed."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles"); os.makedirs(P, exist_ok=True)
summary = {}
for name in ("prof_kt", "prof_kt_serial"):
    f = glob.glob(os.path.join(G, name, "*", "*_kernel_stats.csv"))
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    keep = [r for r in rows if "svo::" in r["Name"]]
    with open(os.path.join(P, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    summary[name] = {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3} for r in keep}
    log = os.path.join(G, name + ".log")
    if os.path.exists(log):
        last = [l for l in open(log).read().splitlines() if l.startswith("{")]
        if last: summary[name + "_bench_line"] = json.loads(last[-1])
pmc = {}
for d in sorted(glob.glob(os.path.join(G, "prof_pmc*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_trace_stack" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
        for k, v in agg.items(): pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
        if agg: pmc["_dispatch"] = meta
summary["pmc_k_trace_stack"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch_kb, write_kb = pmc["FETCH_SIZE"]["mean_per_launch"], pmc["WRITE_SIZE"]["mean_per_launch"]
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B
    # requests at 64 B -> double the read side.  (Uncalibrated for dword gathers: an upper estimate.)
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    summary["hbm_traffic"] = {"fetch_size_raw": fetch_kb, "write_size_raw": write_kb, "hbm_bytes_per_launch_corrected": hbm,
                              "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 FETCH_SIZE half-count"}
    if "TCC_HIT_sum" in pmc:
        summary["l2_hit_rate"] = pmc["TCC_HIT_sum"]["mean_per_launch"] / (pmc["TCC_HIT_sum"]["mean_per_launch"] + pmc["TCC_MISS_sum"]["mean_per_launch"])
    wl = summary.get("prof_kt_bench_line", {}).get("config", {}).get("workload", "c3_1080p_depth12_4x1x4_shadow")
    rec = {"hbm_bytes_per_frame": hbm, "source": f"profiles/{tag}_summary.json (PMC passes: one frame per launch)"}
    if "SQ_INSTS_VALU" in pmc: rec["valu_insts_per_frame"] = pmc["SQ_INSTS_VALU"]["mean_per_launch"]
    json.dump({wl: rec}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if not k.endswith("bench_line")}, indent=1)[:3000])
