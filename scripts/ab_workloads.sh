#!/bin/bash
# variants x workloads, interleaved: usage (on the box): bash scripts/ab_workloads.sh "w1 w2 ..." name1 name2 ...
R=$GRAFT_REPO_ROOT; cd $R
wl="$1"; shift
for rep in 1 2; do for w in $wl; do for name in "$@"; do
  SVO_AMD_LIB=$R/octree-raymarcher_amd/build/libsvo_$name.so timeout -k 10 200 python bench.py --workload $w --no-diagnostics --no-cpu-baseline > gpurun_out/abw.json 2> gpurun_out/abw.err || { echo "$w $name FAILED"; continue; }
  python - "$w" "$name" <<'PY'
import json,sys
r=json.loads(open("gpurun_out/abw.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1][:22]:22s} {sys.argv[2]:8s} {r['value']:9.1f} Mrays/s  kernel_ms={r['roofline']['kernel_ms_avg']}")
PY
done; done; done
