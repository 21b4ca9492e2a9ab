# experiments: A/B of library variants on the single-frame leg (and the run's throughput); usage: bash scripts/ab_single.sh name1 name2 ...
for name in "$@"; do
  SVO_AMD_LIB=$GRAFT_REPO_ROOT/octree-raymarcher_amd/build/libsvo_$name.so python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d.get('single_frame_ms',{}); print('$name: value', d['value'], 'serialized launch ms', d['roofline']['kernel_ms_avg'], 'single frame', s.get('plain_ms'))"
done
