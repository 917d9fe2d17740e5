import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob("gpurun_out/kstat/*kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
sel = [r for r in rows if re.search(sys.argv[1], r["Name"])]
print("all kernels %.3f ms/step; selected (%s) %.3f ms/step in %d launches/step" % (tot, sys.argv[1], sum(float(r["TotalDurationNs"]) for r in sel) / 6e6, sum(int(r["Calls"]) for r in sel) / 6))
for r in sorted(sel, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("   %-90s n/step %5.1f  %7.3f ms" % (r["Name"][:90], int(r["Calls"]) / 6, float(r["TotalDurationNs"]) / 6e6))
