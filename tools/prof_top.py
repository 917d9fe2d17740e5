#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 kernel_stats.csv: tools/prof_top.py <dir> [--steps K --warmup W ...]"""
import csv
import glob
import sys

d = sys.argv[1]
steps, warm = 3, 2
a = sys.argv[2:]
for i, v in enumerate(a):
    if v == "--steps":
        steps = int(a[i + 1])
    if v == "--warmup":
        warm = int(a[i + 1])
n = steps + warm
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernel time per step %.2f ms, launches per step %d" % (tot / n / 1e6, calls / n))
for r in rows[:38]:
    print("%7.3f ms/step %6.1f calls/step %9.1f us avg  %s" % (float(r["TotalDurationNs"]) / n / 1e6, int(r["Calls"]) / n,
                                                                 float(r["AverageNs"]) / 1e3, r["Name"][:90]))
