#!/bin/bash
# Round 4, VERDICT r3 item 1: what would folding the chunk step / hit resolve / refill into the statement buy at most?  The C++
# blocks already serve waiting lanes between statements, by vote; serving EVERY waiting lane after EVERY statement (vote
# thresholds of 1) is the upper bound of what in-statement service can win back in lanes - and shows its price in instructions.
# Variants (scripts/build_variants.sh): base (votes 8 / 16 / refill 8), mid (3 / 6 / 4), eager (1 / 1 / 1); *t = timing builds.
# Per variant: throughput (bench.py, interleaved twice), marching lanes per asm step (scripts/wave_timeline.py, 8 path frames per
# launch), VALU lane utilisation and instruction counts (one --pmc pass).  Results: gpurun_out/serve_always_*.txt
R=$GRAFT_REPO_ROOT; cd $R
AB_REPS=2 bash scripts/ab.sh "--no-diagnostics" base mid eager > gpurun_out/serve_always_ab.txt 2>&1
for v in timing midt eagert; do
  SVO_AMD_LIB=$R/octree-raymarcher_amd/build/libsvo_$v.so SVO_PATH_CAMS=0 python scripts/wave_timeline.py 12 8 2>&1 | tail -8 > gpurun_out/serve_always_lanes_$v.txt
done
for v in base mid eager; do
  SVO_AMD_LIB=$R/octree-raymarcher_amd/build/libsvo_$v.so PMCQ_ARGS="--frames-per-launch 16" bash scripts/pmc_quick.sh > gpurun_out/serve_always_pmc_$v.txt 2>&1
done
cat gpurun_out/serve_always_ab.txt; tail -4 gpurun_out/serve_always_lanes_*.txt; tail -3 gpurun_out/serve_always_pmc_*.txt
