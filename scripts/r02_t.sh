#!/bin/bash
set -e
for cfg in "4 4" "8 2" "8 4" "16 1" "16 2" "16 4" "2 8"; do
  set -- $cfg
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-diagnostics --steps 256 --warmup 32 --frames-per-launch $1 --streams $2 | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F=$1 S=$2', r['value'], 'kernel_ms', r['roofline']['kernel_ms_avg'], 'frac', r['roofline']['frac'])"
done
