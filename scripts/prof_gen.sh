#!/bin/bash
# kernel trace of the device world build (scripts/gen_timing.py: the C3 world twice; the second call is the warm one)
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_gen
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_gen -o gen -- python3 $R/scripts/gen_timing.py > $R/gpurun_out/prof_gen.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os, re
R = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_gen"
f = glob.glob(R + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)
split = gaps[0][1] + 1
second = rows[split:]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in second:
    n = re.sub(r"\(.*", "", n)
    n = "rocprim scan" if "rocprim" in n else n.split("<")[0]
    agg[n][0] += 1; agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
print("warm build: %d dispatches over %.1f ms, kernels busy %.1f ms" % (len(second), (second[-1][1] - second[0][0]) / 1e6, tot / 1e6))
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print("%8.3f ms %6d  %s" % (t / 1e6, c, n[:100]))
PY
