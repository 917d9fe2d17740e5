#!/bin/bash
# usage (GPU box, repo root): tools/kstat.sh <pattern> [bench args]  -> per-launch averages of the kernels matching <pattern> in a short bench run
R=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift
export TMPDIR=/tmp
cd /tmp
rm -rf "$R/gpurun_out/kstat"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/kstat" -o k -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-events "$@" > "$R/gpurun_out/kstat.log" 2>&1
cd "$R"
tail -1 gpurun_out/kstat.log | cut -c1-200
python3 - "$pat" <<'PY'
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob("gpurun_out/kstat/*kernel_stats.csv")[0])))
for r in rows:
    if re.search(sys.argv[1], r["Name"]):
        print("%-70s calls %5d  avg %9.1f us  total/step %8.3f ms" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e6))
PY
