#!/bin/bash
# The driver's command (bench.py --steps 20 --warmup 5) under different launch shapes, interleaved twice on one box.
R=${GRAFT_REPO_ROOT:-.}; cd $R
for rep in 1 2; do
for shape in "2 10" "2 16" "1 16" "3 7" "2 12" "4 5" "1 10"; do
  set -- $shape
  python bench.py --steps 20 --warmup 5 --streams $1 --frames-per-launch $2 --no-diagnostics --no-cpu-baseline > gpurun_out/dshape.json 2> gpurun_out/dshape.err || { echo "S=$1 G=$2 FAILED"; tail -2 gpurun_out/dshape.err; continue; }
  python - "$1" "$2" <<'PY'
import json,sys
r=json.loads(open("gpurun_out/dshape.json").read().strip().splitlines()[-1])
print(f"S={sys.argv[1]} G={sys.argv[2]:>2s}  {r['value']:8.1f} Mrays/s  {r['ms_per_step']:.4f} ms/step")
PY
done
done
