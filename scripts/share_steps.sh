# a rank's 1/N share with the bench defaults, at several run lengths (pipeline ramp-up/-down vs steady state)
for sh in 2 4 8; do for a in "--steps 50 --warmup 5" "--steps 200 --warmup 20" "--steps 800 --warmup 64"; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --emulate-share $sh > gpurun_out/ss.json 2>/dev/null || { echo FAIL; continue; }
  python - $sh <<'PY'
import sys,json
r=json.loads(open("gpurun_out/ss.json").read().strip().splitlines()[-1])
print("share 1/%s steps %4d: %8.1f Mrays/s %.4f ms/frame  S=%d F=%d" % (sys.argv[1], r["steps"], r["value"], r["ms_per_step"], r["config"]["launches_in_flight"], r["config"]["frames_per_launch"]))
PY
done; done
