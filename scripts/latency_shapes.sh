#!/bin/bash
# Launch shapes next to the default one (run on the GPU box): one frame per launch on the path and on the SURVEY camera,
# four frames per launch serialized (round 1's roofline shape).  Prints value / ms per frame / kernel ms per launch.
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-diagnostics --steps 64 --warmup 32 "$@" 2>gpurun_out/lat.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'Mrays/s', r['ms_per_step'], 'ms/frame  kernel_ms', r['roofline']['kernel_ms_avg'], 'frac', r['roofline']['frac'])" || tail -3 gpurun_out/lat.err; }
run --streams 1 --frames-per-launch 1
run --streams 1 --frames-per-launch 1 --camera-path fixed
run --streams 1 --frames-per-launch 4
run --streams 1 --frames-per-launch 4 --camera-path fixed
run --streams 4 --frames-per-launch 4
