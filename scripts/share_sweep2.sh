for q in 8 16 32; do for cfg in "8 16" "8 32" "8 15" "1 3" "1 8" "1 4" "2 8" "4 16"; do set -- $cfg; share=$1; s=$2
  extra=""; [ "$share" != "1" ] && extra="--emulate-share $share"
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 200 --warmup 10 $extra --streams $s > gpurun_out/share.json 2>gpurun_out/share.err || { echo fail; tail -3 gpurun_out/share.err; }
  python - <<PY
import json
r=json.loads(open("gpurun_out/share.json").read().strip().splitlines()[-1])
print("hwq $q share $share streams $s: %.1f Mrays/s  %.4f ms/step" % (r["value"], r["ms_per_step"]))
PY
done; done
