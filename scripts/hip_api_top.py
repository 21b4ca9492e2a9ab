"""Diagnostic: the longest HIP API calls of a rocprofv3 --hip-trace csv (first-use costs show up here)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
big = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:25]
for r in sorted(big, key=lambda r: int(r["Start_Timestamp"])):
    print("%9.2f ms  +%8.2f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Function"]))
agg = {}
for r in rows:
    a = agg.setdefault(r["Function"], [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("--- totals")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-40s calls %6d  total %8.2f ms" % (k, v[0], v[1] / 1e6))
