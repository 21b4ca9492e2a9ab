#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 scripts/fuzz_parity.py 3 200000 2>&1 | tail -1
timeout -k 10 500 python3 bench.py --no-cpu-baseline > gpurun_out/r02/bench_a.json 2> gpurun_out/r02/bench_a.err || { tail -20 gpurun_out/r02/bench_a.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r02/bench_a.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['diagnostics'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'])
PY
SVO_PATH_CAM=7 SVO_AMD_LIB=$GRAFT_REPO_ROOT/octree-raymarcher_amd/build/libsvo_timing.so timeout -k 10 300 python3 scripts/wave_timeline.py 12 1 2>&1 | tail -30
