# experiments: launches in flight (S) x frames per launch (F): whole-run throughput and the serialized launch behind roofline.frac
# usage: bash scripts/sweep_frames.sh "<bench args>" "S F" "S F" ...
args="$1"; shift
for cfg in "$@"; do
  set -- $cfg
  python bench.py $args --no-diagnostics --no-cpu-baseline --streams $1 --frames-per-launch $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('S=$1 F=$2 -> frames per launch', r['frames_per_launch'], 'value', d['value'], 'launch ms', r['kernel_ms_avg'], 'frac', r['frac'])"
done
