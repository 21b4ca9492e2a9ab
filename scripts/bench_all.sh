#!/bin/bash
# Every bench line of the round (run on the GPU box): the default line, the driver's shape, the other BASELINE configs.
# usage: bash scripts/bench_all.sh  ->  gpurun_out/bench_*.json (copy into profiles/<round>_bench_*.json)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > gpurun_out/bench_c3_final.json 2> gpurun_out/bench_c3_final.err
echo c3 done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_c3_driver_shape.json 2> gpurun_out/bench_c3_driver_shape.err
echo driver shape done
for w in c2_1080p_depth10_1chunk c3_grazing_1080p_depth12_4x1x4_shadow c5_1080p_depth16_sparse_shadow c4_2160p_depth12_4x1x4_shadow; do
  timeout -k 10 500 python3 bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  echo $w done
done
