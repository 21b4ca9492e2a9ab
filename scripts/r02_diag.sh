#!/bin/bash
set -e
export SVO_PATH_CAM=${1:-7}
timeout -k 10 300 python3 scripts/steps_tail.py 2>&1 | tail -30
SVO_AMD_LIB=$GRAFT_REPO_ROOT/octree-raymarcher_amd/build/libsvo_timing.so timeout -k 10 300 python3 scripts/wave_timeline.py 12 1 2>&1 | tail -32
