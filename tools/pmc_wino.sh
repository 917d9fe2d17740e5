#!/bin/bash
# usage (GPU box): tools/pmc_wino.sh   -> SQ counters of the Winograd kernels on the micro-benchmark shapes (gpurun_out/pmc_wino*.csv)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$R/gpurun_out/pmc_wino" -o p -- python3 "$R/tools/bench_wino.py" > "$R/gpurun_out/pmc_wino.log" 2>&1
cd "$R"
python3 - <<'PY'
import csv, collections, glob
f = glob.glob("gpurun_out/pmc_wino/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    if "wino" not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
        print("   %-24s %6.1f %% of wave cycles" % (n, 100 * v.get(n, 0) / wc))
    print("   MFMA busy / busy cycles: %.3f" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1)))
PY
