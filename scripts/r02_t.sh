#!/bin/bash
set -e
V="base slim slim2"
export AB_REPS=2
echo "== fixed pipelined"; bash scripts/ab.sh "--steps 200 --warmup 20 --no-diagnostics --camera-path fixed" $V
echo "== path pipelined"; bash scripts/ab.sh "--steps 200 --warmup 20 --no-diagnostics" $V
