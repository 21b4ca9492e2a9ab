#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 500 python3 bench.py > gpurun_out/r02/bench_b.json 2> gpurun_out/r02/bench_b.err || { tail -20 gpurun_out/r02/bench_b.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r02/bench_b.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['diagnostics']['identical_view_mrays'], {k:v for k,v in r['diagnostics']['per_camera_serialized_mrays'].items() if k!='how'}, r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['cpu_baseline'], r['config']['world_generate_s'])
PY
timeout -k 10 300 python3 bench.py --workload c5_1080p_depth16_sparse_shadow --no-cpu-baseline --no-diagnostics | cut -c1-400
timeout -k 10 300 python3 bench.py --workload c2_1080p_depth10_1chunk --no-cpu-baseline --no-diagnostics | cut -c1-300
