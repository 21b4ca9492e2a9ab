#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 500 python3 bench.py > gpurun_out/r02/bench_c.json 2> gpurun_out/r02/bench_c.err || { tail -20 gpurun_out/r02/bench_c.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r02/bench_c.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['diagnostics']['identical_view_mrays'], {k:v for k,v in r['diagnostics']['per_camera_serialized_mrays'].items() if k not in('how','all')}, r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['roofline']['frac_throughput'], r['cpu_baseline']['parity_with_gpu'], r['cpu_baseline']['value'], r['cpu_baseline']['single_thread']['value'], r['roofline']['valu_issue'])
PY
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_driver_shape.json; python3 -c "import json,sys; r=json.loads(open('gpurun_out/r02/bench_driver_shape.json').read().strip().splitlines()[-1]); print('driver shape', r['value'], r['ms_per_step'], r['roofline']['frac'], r['config']['frames_per_launch'], r['config']['launches_in_flight'])"
for wl in c2_1080p_depth10_1chunk c4_2160p_depth12_4x1x4_shadow c5_1080p_depth16_sparse_shadow c3_grazing_1080p_depth12_4x1x4_shadow; do
timeout -k 10 400 python3 bench.py --workload $wl > gpurun_out/r02/bench_$wl.json 2>/dev/null; python3 -c "
import json,sys; r=json.loads(open('gpurun_out/r02/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['frac_throughput'], r['cpu_baseline']['value'], r['cpu_baseline']['parity_with_gpu'], r['config']['rays_per_frame'])"
done
PMCQ_ARGS="" bash scripts/pmc_quick.sh 2>&1 | tail -12
