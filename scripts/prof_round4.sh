#!/bin/bash
# Round-4 profiling recipe (run on the GPU box through gpurun).  Kernel trace + separate PMC passes (never combined with
# trace domains other than --kernel-trace); summaries are copied to profiles/ afterwards by scripts/summarize_profiles.py r04.
# New this round: the PMC passes run on the REPORTED launch shape (16 frames per launch), serialized (--streams 1) and with
# two launches in flight, next to the one-frame-per-launch passes that price traffic per frame.
set -e
set -x
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_kt_serial $R/gpurun_out/prof_pmc* $R/gpurun_out/prof_*.log
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-diagnostics"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- $BENCH --steps 64 --warmup 32 > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt_serial -- $BENCH --steps 64 --warmup 32 --streams 1 > $R/gpurun_out/prof_kt_serial.log 2>&1
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"
C2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
# (a) the reported shape, serialized: 16 frames per launch, one launch at a time
P="--steps 32 --warmup 16 --streams 1 --frames-per-launch 16"
rocprofv3 --pmc $C1 --output-format csv -d $R/gpurun_out/prof_pmcA1 -- $BENCH $P > $R/gpurun_out/prof_pmcA1.log 2>&1
rocprofv3 --pmc $C2 --output-format csv -d $R/gpurun_out/prof_pmcA2 -- $BENCH $P > $R/gpurun_out/prof_pmcA2.log 2>&1
# (b) the reported shape, two launches in flight
P="--steps 64 --warmup 32 --streams 2 --frames-per-launch 16"
rocprofv3 --pmc $C1 --output-format csv -d $R/gpurun_out/prof_pmcB1 -- $BENCH $P > $R/gpurun_out/prof_pmcB1.log 2>&1
rocprofv3 --pmc $C2 --output-format csv -d $R/gpurun_out/prof_pmcB2 -- $BENCH $P > $R/gpurun_out/prof_pmcB2.log 2>&1
# (c) one frame per launch, every camera of the path once: traffic and instruction counts per frame
P="--steps 32 --warmup 1 --streams 1 --frames-per-launch 1"
rocprofv3 --pmc $C1 --output-format csv -d $R/gpurun_out/prof_pmc1 -- $BENCH $P > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/prof_pmc3 -- $BENCH $P > $R/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc4 -- $BENCH $P > $R/gpurun_out/prof_pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_pmc5 -- $BENCH $P > $R/gpurun_out/prof_pmc5.log 2>&1
# condense on the box (gpurun copies back at most 64 MiB; a counter CSV has one row per dispatch, counter and instance)
python3 - <<'PY'
import csv, glob, collections, json, os
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = {}
for d in sorted(glob.glob(R + "/prof_pmc*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
        for r in csv.DictReader(open(f)):
            if "k_trace_stack" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
        # launches of the shape under study only: the largest grid (the preamble's one-frame counter launches are smaller or literal)
        res = {}
        for c, per in agg.items():
            v = sorted(per.values())
            res[c] = {"mean_per_launch": sum(v) / len(v), "median_per_launch": v[len(v) // 2], "launches": len(v)}
        out[os.path.basename(d)] = {"counters": res, "dispatch": meta}
json.dump(out, open(R + "/prof_pmc_r04.json", "w"), indent=1)
print(json.dumps({k: {c: round(v["median_per_launch"]) for c, v in d["counters"].items()} for k, d in out.items()}, indent=0))
PY
rm -rf $R/gpurun_out/prof_pmc*/ ; find $R/gpurun_out -name "*_kernel_trace.csv" -delete; find $R/gpurun_out -name "*_agent_info.csv" -delete; du -sh $R/gpurun_out
tail -c 600 $R/gpurun_out/prof_kt.log; echo
tail -c 600 $R/gpurun_out/prof_kt_serial.log; echo
