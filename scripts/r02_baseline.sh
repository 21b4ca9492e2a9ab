#!/bin/bash
# round-2 opening measurements: VALU issue microbenchmark + the round-1 kernel on and off the lattice plane
set -e
mkdir -p gpurun_out/r02
./scripts/microbench/valu_issue > gpurun_out/r02/valu_issue.json
cat gpurun_out/r02/valu_issue.json | head -40
python3 bench.py --steps 48 --warmup 8 --streams 1 --no-cpu-baseline > gpurun_out/r02/base_on_serial.json
SVO_BENCH_EYE_DX=0.31 python3 bench.py --steps 48 --warmup 8 --streams 1 --no-cpu-baseline > gpurun_out/r02/base_off_serial.json
SVO_BENCH_EYE_DX=0.31 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/base_off_pipe.json
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/base_on_pipe.json
tail -n 1 gpurun_out/r02/base_*.json
