#!/bin/bash
# Stage counters of the sure-miss brick test (step_asm_body.inc): lanes per stage, draining waves only (1..4) and every wave (A1, A2, A4).
# Needs build/libsvo_surestat{1,2,3,4,A1,A2,A4}.so (-DSVO_STACK_TIMING -DSVO_SURE_STAT_WORD -DSVO_SURE_STAT=k [-DSVO_SURE_MISS_WHEN=1]).
cd "$(dirname "$0")/.."
for k in ${SURE_STAGES:-1 2 3 4 A1 A2 A4}; do
  echo "== stage $k"
  SVO_AMD_LIB=octree-raymarcher_amd/build/libsvo_surestat$k.so SVO_PATH_CAM=${SVO_PATH_CAM:-12} python scripts/wave_timeline.py 12 1 | grep "kernel span\|^asm steps\|lane-steps total\|avg lanes per iteration"
done
