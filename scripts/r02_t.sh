#!/bin/bash
set -e
V="base p100 p200 p400"
export AB_REPS=2
echo "== path pipelined"; bash scripts/ab.sh "--steps 200 --warmup 24 --no-diagnostics" $V
echo "== single frame serial"; bash scripts/ab.sh "--steps 32 --warmup 32 --streams 1 --frames-per-launch 1 --no-diagnostics" $V
