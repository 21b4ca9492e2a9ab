#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun).  Kernel trace + separate PMC passes
# (never combined with trace domains other than --kernel-trace), summaries copied to profiles/ afterwards
# by scripts/summarize_profiles.py r02.
set -e
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_kt_serial $R/gpurun_out/prof_pmc* $R/gpurun_out/prof_calib* $R/gpurun_out/prof_*.log $R/gpurun_out/prof_valu_issue.json
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-diagnostics"
# the timed command of the round: 32-camera path, 4 frames per launch, 4 launches in flight / serialized
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- $BENCH --steps 64 --warmup 32 > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt_serial -- $BENCH --steps 64 --warmup 32 --streams 1 > $R/gpurun_out/prof_kt_serial.log 2>&1
# counters per frame: one frame per launch, every camera of the path once in the timed region (and once each in the untimed
# preamble): the mean over k_trace_stack dispatches is the mean over the path's frames
P="--steps 32 --warmup 1 --streams 1 --frames-per-launch 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_pmc1 -- $BENCH $P > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/prof_pmc2 -- $BENCH $P > $R/gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/prof_pmc3 -- $BENCH $P > $R/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc4 -- $BENCH $P > $R/gpurun_out/prof_pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_pmc5 -- $BENCH $P > $R/gpurun_out/prof_pmc5.log 2>&1
# what FETCH_SIZE makes of dword / qword gathers and of a coalesced stream (known byte counts)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_calib1 -- $R/scripts/microbench/gather_calib > $R/gpurun_out/prof_calib.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_calib2 -- $R/scripts/microbench/gather_calib >> $R/gpurun_out/prof_calib.log 2>&1
# VALU issue microbenchmark under the kernel trace (its JSON is the result; the trace shows the launches)
$R/scripts/microbench/valu_issue > $R/gpurun_out/prof_valu_issue.json
tail -1 $R/gpurun_out/prof_kt.log
tail -1 $R/gpurun_out/prof_kt_serial.log
tail -2 $R/gpurun_out/prof_calib.log
