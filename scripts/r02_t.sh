#!/bin/bash
set -e
timeout -k 10 900 python3 -m pytest tests/test_shading.py -m gpu -x -q 2>&1 | tail -3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_aux
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_aux -- python3 $R/scripts/aux_kernels_prof.py > $R/gpurun_out/prof_aux.log 2>&1
grep -h "k_shade\|k_gbuffer\|k_brick_masks" $R/gpurun_out/prof_aux/*/*_kernel_stats.csv | cut -c1-160
