#!/bin/bash
# pool-kernel policy variants: single frame time, lanes at a statement's start (scripts/pool_debug.py)
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  echo "== $v"; SVO_AMD_LIB=$R/octree-raymarcher_amd/build/libsvo_$v.so timeout -k 10 90 python scripts/pool_debug.py 12 1920 1080 1 2>&1 | grep -E "stack kernel|marching lanes|frames:|inplace|idle_spins|serve_|ticks|iters"
done
