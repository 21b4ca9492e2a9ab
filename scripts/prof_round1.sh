#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun).  Kernel trace + separate PMC passes
# (never combined with trace domains other than --kernel-trace), summaries copied to profiles/ afterwards.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- $BENCH --steps 48 --warmup 4 > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt_serial -- $BENCH --steps 48 --warmup 4 --streams 1 > $R/gpurun_out/prof_kt_serial.log 2>&1
P="--steps 4 --warmup 1 --streams 1 --frames-per-launch 1"      # counters per frame: one frame per launch
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_pmc1 -- $BENCH $P > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/prof_pmc2 -- $BENCH $P > $R/gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/prof_pmc3 -- $BENCH $P > $R/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc4 -- $BENCH $P > $R/gpurun_out/prof_pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_pmc5 -- $BENCH $P > $R/gpurun_out/prof_pmc5.log 2>&1
tail -1 $R/gpurun_out/prof_kt.log
tail -1 $R/gpurun_out/prof_kt_serial.log
