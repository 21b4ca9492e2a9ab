#!/bin/bash
# Run bench.py once per variant library on the GPU box (AB_REPS times, variants interleaved: A B A B ...);
# prints value / ms_per_step per variant and run.
# usage (on the box): [AB_REPS=2] bash scripts/ab.sh "<bench args>" name1 name2 ...
args="$1"; shift
for rep in $(seq 1 ${AB_REPS:-1}); do
for name in "$@"; do
  SVO_AMD_LIB=$GRAFT_REPO_ROOT/octree-raymarcher_amd/build/libsvo_$name.so timeout -k 10 300 python bench.py $args --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/ab_$name.err; continue; }
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
r=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
print(f"{n:14s} {r['value']:10.1f} Mrays/s  {r['ms_per_step']:.4f} ms/step  kernel_ms={r.get('roofline',{}).get('kernel_ms_avg')}")
PY
done
done
