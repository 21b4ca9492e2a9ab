#!/bin/bash
set -e
V="base again2"
export AB_REPS=2
echo "== path pipelined"; bash scripts/ab.sh "--steps 200 --warmup 20 --no-diagnostics" $V
echo "== path serial"; bash scripts/ab.sh "--steps 64 --warmup 32 --streams 1 --no-diagnostics" $V
