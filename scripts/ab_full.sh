#!/bin/bash
# Like ab.sh, but prints the single-frame figures of the bench line as well (run without --no-diagnostics for them).
# usage (on the box): [AB_REPS=2] bash scripts/ab_full.sh "<bench args>" name1 name2 ...
args="$1"; shift
for rep in $(seq 1 ${AB_REPS:-1}); do
for name in "$@"; do
  SVO_AMD_LIB=$GRAFT_REPO_ROOT/octree-raymarcher_amd/build/libsvo_$name.so timeout -k 10 400 python bench.py $args --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/ab_$name.err; continue; }
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
r=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
sf=r.get('single_frame_ms') or {}
pl=sf.get('plain_ms',{}); od=sf.get('ordered_ms',{})
print(f"{n:10s} {r['value']:8.1f} Mrays/s {r['ms_per_step']:.4f} ms/step kernel_ms={r.get('roofline',{}).get('kernel_ms_avg')} single plain {pl.get('mean')}/{pl.get('max')} ordered {od.get('mean')}/{od.get('max')}")
PY
done
done
