# Per-rank kernel work of an N-GPU run, emulated on one GPU: rank 0's 1/N share of the C3 frame, S frames in flight.
for cfg in "1 8" "2 8" "2 16" "4 16" "4 32" "8 16" "8 32" "8 48"; do set -- $cfg; share=$1; s=$2
  extra=""; [ "$share" != "1" ] && extra="--emulate-share $share"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 400 --warmup 20 $extra --streams $s > gpurun_out/share.json 2>gpurun_out/share.err || { echo fail; tail -3 gpurun_out/share.err; continue; }
  python - <<PY
import json
r=json.loads(open("gpurun_out/share.json").read().strip().splitlines()[-1])
print("share 1/$share streams $s: %.1f Mrays/s  %.4f ms/step  -> x$share = %.0f Mrays/s aggregate if it scaled" % (r["value"], r["ms_per_step"], r["value"]*$share))
PY
done
