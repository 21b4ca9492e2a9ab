# Round-end records: the bench line of every BASELINE config (+ C3 in the driver's shape) into gpurun_out/bench_<name>.json
set -x
R=${GRAFT_REPO_ROOT:-.}
cd $R
python bench.py > gpurun_out/bench_c3_final.json 2> gpurun_out/bench_c3_final.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c3_driver_shape.json 2> gpurun_out/bench_c3_driver_shape.err
for w in c2_1080p_depth10_1chunk c3_grazing_1080p_depth12_4x1x4_shadow c4_2160p_depth12_4x1x4_shadow c5_1080p_depth16_sparse_shadow; do
  python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_c*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("single_frame_ms", {}).get("plain_ms"), d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
