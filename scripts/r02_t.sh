#!/bin/bash
set -e
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
V="base vw4 vw16 vh8 vh32 rf4 rf16 sl8"
export AB_REPS=2
echo "== path pipelined"; bash scripts/ab.sh "--steps 200 --warmup 24 --no-diagnostics" $V
