#!/bin/bash
# A rank's 1/N share of the frame at the DRIVER's run length (--steps 20 --warmup 5), launches in flight x frames per launch:
# what SCALE_rNN would see per rank (trace only, one GPU emulating one rank's bands; the gather is not in it).
for rep in 1 2; do
for sh in 2 4 8; do for shape in "0 0" "2 10" "4 5" "1 16" "2 8" "8 2"; do
  set -- $shape
  extra=""; [ "$1" != "0" ] && extra="--streams $1 --frames-per-launch $2"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-diagnostics --emulate-share $sh $extra > gpurun_out/sd.json 2>/dev/null || { echo FAIL $sh $shape; continue; }
  python - $sh <<'PY'
import sys,json
r=json.loads(open("gpurun_out/sd.json").read().strip().splitlines()[-1])
print("share 1/%s: %8.1f Mrays/s of the share's rays, %.4f ms/frame  S=%d F=%d" % (sys.argv[1], r["value"], r["ms_per_step"], r["config"]["launches_in_flight"], r["config"]["frames_per_launch"]))
PY
done; done; done
