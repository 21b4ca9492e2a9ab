#!/bin/bash
set -e
mkdir -p gpurun_out/r02
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 scripts/fuzz_parity.py 2 200000 2>&1 | tail -2
V="nocreep s6 s3"
echo "== on-lattice serial"; bash scripts/ab.sh "--steps 48 --warmup 8 --streams 1" $V
echo "== off-lattice serial"; SVO_BENCH_EYE_DX=0.31 bash scripts/ab.sh "--steps 48 --warmup 8 --streams 1" $V
echo "== on-lattice pipelined"; bash scripts/ab.sh "--steps 200 --warmup 20" $V
echo "== off-lattice pipelined"; SVO_BENCH_EYE_DX=0.31 bash scripts/ab.sh "--steps 200 --warmup 20" $V
SVO_BENCH_EYE_DX=0.31 SVO_AMD_LIB=$GRAFT_REPO_ROOT/octree-raymarcher_amd/build/libsvo_timing.so timeout -k 10 300 python3 scripts/wave_timeline.py 12 1 2>&1 | tail -28
