"""Condense gpurun_out/prof_* of scripts/prof_round4.sh (+ scripts/microbench/valu_issue) into the committed profiles/r04_*
files and profiles/traffic.json (what bench.py replays, labelled as replayed).  Runs anywhere."""
import collections, csv, glob, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles"); tag = "r04"
S = {}
for name in ("prof_kt", "prof_kt_serial"):
    f = sorted(glob.glob(os.path.join(G, name, "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)     # (gpurun merges into gpurun_out/: older runs' files stay)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    keep = [r for r in rows if "svo::" in r["Name"]]
    with open(os.path.join(P, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    S[name] = {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3} for r in keep}
    last = [l for l in open(os.path.join(G, name + ".log")).read().splitlines() if l.startswith("{")]
    if last: S[name + "_bench_line"] = json.loads(last[-1])
pmc = json.load(open(os.path.join(G, "prof_pmc_r04.json")))
S["pmc"] = {"how": "separate rocprofv3 --pmc passes (scripts/prof_round4.sh); per k_trace_stack launch, median over the launches of the pass. "
                   "A: 16 frames per launch, serialized.  B: 16 frames per launch, two launches in flight.  1/3/4/5: one frame per launch over the 32-camera path",
            "passes": pmc}
def med(p, c): return pmc[p]["counters"][c]["median_per_launch"]
# ---- instruction issue: ONE figure with its spread
# the instruction prices are the hardware's: the committed microbenchmark run of round 3 (profiles/r03_valu_issue.json) unless this round re-ran it
vi_path = os.path.join(G, "prof_valu_issue.json") if os.path.exists(os.path.join(G, "prof_valu_issue_r04.marker")) else os.path.join(P, "r03_valu_issue.json")
vi = json.load(open(vi_path))
def cost(op, w, idx=1):
    for r in vi["runs"]:
        if r["op"] == op and r["waves_per_simd"] == w and r["exec"] == "all 64 lanes": return r["cycles_per_inst_per_simd_grouped_p10_median_p90"][idx]
src = open(os.path.join(ROOT, "octree-raymarcher_amd", "csrc", "step_asm_body.inc")).read()
body = src[src.index("asm volatile("):src.index(': [md] "+v"(mode)')]
body = body[:body.index('"s_cbranch_scc0 60f')] + '"s_cbranch_scc0 x\\n"\n' + body[body.index('"60:'):]      # (the sure-miss test of draining waves is not part of a bulk step's mix)
body = body.replace("SVO_STEP_LOAD_ENTRY", '"v_lshl_add_u32 x\\n" "global_load_dword x\\n"').replace("SVO_STEP_MASK_OFFSET", '"v_lshlrev_b32 x\\n" "s_nop 0\\n"') \
           .replace("SVO_STEP_LOAD_MASK", '"global_load_dwordx2 x\\n"').replace("SVO_STEP_LEAF_DISTANCE", '"v_subrev_f32 x\\n"').replace("SVO_STEP_ESCAPE_GUARD", "")     # the default variant's fragments (step_asm.hip.h)
ins = [i for i in re.findall(r'"\s*([a-z_0-9]+)[ \\]', body) if not i[0].isdigit()]
mix = collections.Counter()
for i in ins:
    if i.startswith("v_cmp") or i == "v_cndmask_b32" or i.startswith("v_subbrev"): mix["v_cmp_cndmask"] += 1
    elif i in ("v_bfe_u32", "v_med3_i32", "v_or3_b32", "v_lshl_add_u32", "v_lshl_or_b32", "v_max3_i32", "v_min3_f32", "v_bfm_b32", "v_lshlrev_b64"): mix["v_three_operand_int"] += 1
    elif i.startswith("v_"): mix["v_plain"] += 1
    elif "branch" in i: mix["branch"] += 1
    elif i in ("s_nop", "s_waitcnt"): mix["s_nop_waitcnt"] += 1
    elif i.startswith("s_"): mix["salu"] += 1
    else: mix["memory"] += 1
W = 6
def weighted(idx):
    fma = cost("v_fma_f32", W, idx)
    c = {"v_plain": fma, "v_cmp_cndmask": cost("v_cmp_lt_f32+v_cndmask_b32", W, idx), "v_three_operand_int": cost("v_bfe_u32", W, idx),
         "salu": cost("v_fma_f32 + s_and_b64", W, idx) - fma, "memory": fma,
         "branch": 0.5 * ((cost("v_fma_f32 + s_cbranch_execz (not taken)", W, idx) - fma) + (cost("v_fma_f32 + s_cbranch_execnz (taken, to the next instruction)", W, idx) - fma)),      # (the companions are timed per PAIR)
         "s_nop_waitcnt": max(0.0, cost("v_fma_f32 + s_nop 1", W, idx) - fma)}
    return sum(mix[k] * c[k] for k in mix) / sum(mix.values()), c
c_med, per_class = weighted(1); c_lo = weighted(0)[0]; c_hi = weighted(2)[0]
insts = sum(med("prof_pmcA1", k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS")) + med("prof_pmcA2", "SQ_INSTS_VMEM_WR")
gfx_cycles = med("prof_pmcA2", "GRBM_GUI_ACTIVE") / 8.0              # summed over the 8 XCCs
FPL = 16        # frames per launch of passes A and B (scripts/prof_round4.sh)
issue = {"instructions_per_launch_of_%d_frames" % FPL: insts, "instructions_per_frame": insts / FPL,
         "valu_per_frame": med("prof_pmcA1", "SQ_INSTS_VALU") / FPL, "salu_per_frame": med("prof_pmcA1", "SQ_INSTS_SALU") / FPL,
         "static_mix_of_the_asm_step": dict(mix), "cycles_per_instruction_by_class_at_6_waves": {k: round(v, 3) for k, v in per_class.items()},
         "cycles_per_instruction_weighted": round(c_med, 3), "spread_p10_p90_over_simds": [round(c_lo, 3), round(c_hi, 3)],
         "gfx_cycles_per_launch": gfx_cycles, "simds": 1024,
         "issue_utilisation": round(insts / 1024 * c_med / gfx_cycles, 4),
         "issue_utilisation_spread": [round(insts / 1024 * c_lo / gfx_cycles, 4), round(insts / 1024 * c_hi / gfx_cycles, 4)],
         # the figure above is a MODEL (static mix x microbenchmark prices).  The direct bound from the same PMC passes:
         # instructions per SIMD x [the cheapest class price, the weighted price] / GFX-busy cycles - quote the range
         "issue_utilisation_direct_range": [round(insts / 1024 * per_class["v_plain"] / gfx_cycles, 4), round(insts / 1024 * c_med / gfx_cycles, 4)],
         "wave_wait_fraction": round(med("prof_pmcA2", "SQ_WAIT_ANY") / med("prof_pmcA1", "SQ_WAVE_CYCLES"), 4),
         "wave_wait_inst_fraction": round(med("prof_pmcA2", "SQ_WAIT_INST_ANY") / med("prof_pmcA1", "SQ_WAVE_CYCLES"), 4),
         "wave_active_fraction": round(med("prof_pmcA1", "SQ_ACTIVE_INST_ANY") / med("prof_pmcA1", "SQ_WAVE_CYCLES"), 4),
         "lane_utilisation_valu": round(med("prof_pmcA2", "SQ_THREAD_CYCLES_VALU") / (64 * med("prof_pmcA2", "SQ_ACTIVE_INST_VALU")), 4),
         "note_model_above_one": "the prices are measured on streams of ONE instruction class at six waves; in a mixed stream a scalar instruction of one wave issues beside a vector "
                                 "instruction of another, so the weighted price is an upper bound and a figure above 1 reads as: the SIMDs' issue slots are full",
         "how": "instructions of one serialized 16-frame launch (PMC pass A) / 1024 SIMDs x cycles per instruction, weighted with the static mix of the "
                "hand-written step (step_asm.hip.h) from scripts/microbench/valu_issue at 6 waves per SIMD (waves grouped by the SIMD they ran on, each "
                "SIMD's busy interval at the shader clock measured in the same run) / GFX-busy cycles of the launch (GRBM_GUI_ACTIVE / 8 XCCs)"}
S["issue"] = issue
fetch_kb, write_kb = med("prof_pmc4", "FETCH_SIZE"), med("prof_pmc5", "WRITE_SIZE")
fabric = (2.0 * fetch_kb + write_kb) * 1024.0
S["fabric_traffic_per_frame"] = {"fetch_size_raw": fetch_kb, "write_size_raw": write_kb, "bytes": fabric,
                                 "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: every read request is a 128-B line tallied at 64 B (profiles/r02_gather_calibration.json)"}
S["l2_hit_rate"] = med("prof_pmc3", "TCC_HIT_sum") / (med("prof_pmc3", "TCC_HIT_sum") + med("prof_pmc3", "TCC_MISS_sum"))
json.dump(S, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
rec = {"fabric_bytes_per_frame": fabric, "l2_hit_rate": round(S["l2_hit_rate"], 4), "tcc_miss_per_frame": med("prof_pmc3", "TCC_MISS_sum"),
       "source": "profiles/r04_summary.json (PMC passes of scripts/prof_round4.sh: traffic from one-frame launches over the 32-camera path, instruction issue from serialized 16-frame launches)",
       "issue": {k: issue[k] for k in ("instructions_per_frame", "valu_per_frame", "salu_per_frame", "cycles_per_instruction_weighted", "spread_p10_p90_over_simds",
                                       "issue_utilisation", "issue_utilisation_spread", "issue_utilisation_direct_range", "wave_wait_fraction", "wave_active_fraction", "lane_utilisation_valu")}}
json.dump({"c3_1080p_depth12_4x1x4_shadow": rec}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in S.items() if k not in ("pmc",) and not k.endswith("bench_line")}, indent=1)[:3500])
