for share in 8 4 2; do for s in 3 5 7 9 12 15 24; do
  python bench.py --no-cpu-baseline --steps 200 --warmup 10 --emulate-share $share --streams $s > gpurun_out/share.json 2>gpurun_out/share.err || { echo fail; tail -3 gpurun_out/share.err; }
  python - <<PY
import json
r=json.loads(open("gpurun_out/share.json").read().strip().splitlines()[-1])
print("share $share streams $s: %.1f Mrays/s  %.4f ms/step  -> x$share = %.0f Mrays/s" % (r["value"], r["ms_per_step"], r["value"]*1))
PY
done; done
