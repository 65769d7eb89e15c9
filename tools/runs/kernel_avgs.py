#!/usr/bin/env python3
"""Prints the average duration of the main kernels from a rocprofv3 --stats directory, and the bench line's rate."""
import csv, glob, json, sys
d, line = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
out = []
for r in rows[:6]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    out.append("%s %sx %.3f ms (%.1f %%)" % (name, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
print("  " + "; ".join(out))
try:
    b = json.loads(open(line).read().strip().splitlines()[-1])
    print("  %.1f Msamples/s, %.2f ms per step" % (b["value"], b["ms_per_step"]))
except Exception as e:  # a probe build with wrong results may fail the image check
    print("  no bench line:", e)
