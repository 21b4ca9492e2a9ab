# experiments: single-frame latency against the persistent grid's size (waves per CU; 24 = six per SIMD, the default)
for g in 24 20 16 12 8; do
  SVO_GRID_WAVES_PER_CU=$g python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves/CU $g', d['value'], d.get('single_frame_ms'))"
done
