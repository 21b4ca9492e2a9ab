#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 500 python3 bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err || { tail -20 gpurun_out/r02/bench_default.err; exit 1; }
cat gpurun_out/r02/bench_default.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_driver_shape.json 2>> gpurun_out/r02/bench_default.err
cat gpurun_out/r02/bench_driver_shape.json
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --steps 64 --warmup 16 > gpurun_out/r02/bench_gloo2.json 2> gpurun_out/r02/bench_gloo2.err || { tail -20 gpurun_out/r02/bench_gloo2.err; exit 1; }
cat gpurun_out/r02/bench_gloo2.json
