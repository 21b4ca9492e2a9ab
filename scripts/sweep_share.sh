# experiments: a rank's 1/N share of the frame on one GPU (--emulate-share N): launches in flight x frames per launch
for share in 2 4 8; do
  for cfg in "4 8" "4 16" "2 16" "8 16"; do
    set -- $cfg
    python bench.py --emulate-share $share --steps 256 --warmup 32 --no-diagnostics --no-cpu-baseline --streams $1 --frames-per-launch $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('share 1/$share S=$1 F=$2 value', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
