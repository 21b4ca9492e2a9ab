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
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in units of 1024 B; on gfx950 FETCH_SIZE tallies every memory-side
    # read request at 64 B although a request is one 128-B line - also for single-dword and 8-byte gathers, calibrated in
    # profiles/<tag>_gather_calibration.json (scripts/microbench/gather_calib.hip) -> double the read side.  These are
    # L2-miss fabric bytes: Infinity-Cache hits are counted, so an upper bound on HBM bytes.
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    summary["fabric_traffic"] = {"fetch_size_raw": fetch_kb, "write_size_raw": write_kb, "fabric_bytes_per_launch_corrected": hbm,
                                 "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: every read request is a 128-B line tallied at 64 B (calibrated for dword / qword gathers and streams)"}
    if "TCC_HIT_sum" in pmc:
        summary["l2_hit_rate"] = pmc["TCC_HIT_sum"]["mean_per_launch"] / (pmc["TCC_HIT_sum"]["mean_per_launch"] + pmc["TCC_MISS_sum"]["mean_per_launch"])
    wl = summary.get("prof_kt_bench_line", {}).get("config", {}).get("workload", "c3_1080p_depth12_4x1x4_shadow")
    rec = {"fabric_bytes_per_frame": hbm, "source": f"profiles/{tag}_summary.json (PMC passes: one frame per launch, mean over the 32-camera path)"}
    if "l2_hit_rate" in summary: rec["l2_hit_rate"] = round(summary["l2_hit_rate"], 4)
    if "SQ_INSTS_VALU" in pmc: rec["valu_insts_per_frame"] = pmc["SQ_INSTS_VALU"]["mean_per_launch"]
    vi = os.path.join(G, "prof_valu_issue.json")
    if os.path.exists(vi):
        runs = json.load(open(vi))["runs"]
        # the seven plain instruction mixes with every lane on (the file also holds EXEC patterns and scalar companions)
        plain = {}
        for r in runs:
            if r["waves_per_simd"] == 5 and r.get("exec", "all 64 lanes") == "all 64 lanes" and " + " not in r["op"]:
                plain.setdefault(r["op"], r)
        five = list(plain.values())
        # wall-clock cycles per instruction per SIMD at 5 waves/SIMD (the march kernel's occupancy), mean over the instruction mixes
        key = [k for k in five[0] if k.startswith("cycles_per_inst_per_simd_from_wall")][0]
        rec["valu_cycles_per_inst"] = round(sum(r[key] for r in five) / len(five), 3)
        json.dump(json.load(open(vi)), open(os.path.join(P, f"{tag}_valu_issue.json"), "w"), indent=1)
    json.dump({wl: rec}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
# FETCH_SIZE calibration on known byte counts
cal = {}
for d in ("prof_calib1", "prof_calib2"):
    for f in glob.glob(os.path.join(G, d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("k_", "void k_")):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                cal.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if cal:
    out = {"what": "scripts/microbench/gather_calib.hip: 2^24 distinct 128-B lines of a 4 GiB table touched once per kernel (table >> 256 MiB Infinity Cache)",
           "lines_touched": 1 << 24, "kernels": {}}
    for name, c in cal.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        e = {"counters_mean": m}
        if "FETCH_SIZE" in m: e["fetch_size_bytes_per_line"] = m["FETCH_SIZE"] * 1024.0 / (1 << 24)
        if "TCC_EA0_RDREQ_sum" in m: e["read_requests_per_line"] = m["TCC_EA0_RDREQ_sum"] / (1 << 24)
        out["kernels"][name] = e
    out["conclusion"] = ("one memory-side read request per 128-B line whatever part of it was asked for (one dword, two dwords 64 B apart, "
                         "8 bytes, or all of it as 16-B-per-lane loads), tallied by FETCH_SIZE at 64 B: bytes moved = 2 * FETCH_SIZE * 1024")
    json.dump(out, open(os.path.join(P, f"{tag}_gather_calibration.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if not k.endswith("bench_line")}, indent=1)[:4000])
